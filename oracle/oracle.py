"""ctypes binding of the CPU oracle (oracle/libsd_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])


def build(force=False):
    """make decides staleness (every *.inc / *.h is a prerequisite in the Makefile)."""
    so = os.path.join(_HERE, "libsd_oracle.so")
    if force and os.path.exists(so):
        os.remove(so)
    if os.path.exists(os.path.join(_HERE, "Makefile")):
        try:
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        except (OSError, subprocess.CalledProcessError):
            if not os.path.exists(so):
                raise
    return so


def _load(so):
    L = C.CDLL(so)
    L.orc_extractor_create.restype = C.c_void_p
    L.orc_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    L.orc_extractor_destroy.argtypes = [C.c_void_p]
    L.orc_fast_atan2.restype = C.c_float
    L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    return L


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _load(build())
    return _LIB


def use_native():
    """bench.py's cpu_baseline leg: the same sources built with -march=native on the box that runs them (BASELINE.md section 2;
    the portable libsd_oracle.so is -march=x86-64-v3 because it travels between machines)."""
    global _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
    _LIB = _load(os.path.join(_HERE, "_native", "libsd_oracle_native.so"))
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Extractor:
    """Mirror of ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-114) on numpy buffers."""

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = C.c_void_p(self.L.orc_extractor_create(nfeatures, scale_factor, nlevels, ini_th, min_th))
        self.scale = np.zeros(nlevels, np.float32); self.inv_scale = np.zeros(nlevels, np.float32)
        self.sigma2 = np.zeros(nlevels, np.float32); self.inv_sigma2 = np.zeros(nlevels, np.float32)
        self.quota = np.zeros(nlevels, np.int32); self.umax = np.zeros(16, np.int32)
        self.L.orc_extractor_tables(self.h, _p(self.scale), _p(self.inv_scale), _p(self.sigma2), _p(self.inv_sigma2),
                                    _p(self.quota), _p(self.umax))

    def __del__(self):
        try:
            self.L.orc_extractor_destroy(self.h)
        except Exception:
            pass

    def set_blur_taps(self, taps):
        t = np.asarray(taps, np.uint16)
        assert t.shape == (7,)
        self.L.orc_extractor_set_blur_taps(self.h, _p(t))

    def __call__(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        cap = self.nfeatures + 64 * self.nlevels + 64
        kp = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        self.per_level = np.zeros(self.nlevels, np.int32); self.cand_per_level = np.zeros(self.nlevels, np.int32)
        n = self.L.orc_extract(self.h, _p(gray), w, h, gray.strides[0], _p(kp), _p(desc), cap, _p(self.per_level),
                               _p(self.cand_per_level))
        assert n >= 0, "oracle capacity too small"
        return kp[:n].copy(), desc[:n].copy()

    def pyramid(self, level):
        """Padded plane of mvImagePyramid[level] (19-px REFLECT_101 border) and its interior view."""
        W, H = C.c_int(), C.c_int()
        self.L.orc_pyramid_dims(self.h, level, C.byref(W), C.byref(H))
        buf = np.zeros((H.value + 38, W.value + 38), np.uint8)
        self.L.orc_pyramid_copy(self.h, level, _p(buf))
        return buf

    def blurred(self, level):
        W, H = C.c_int(), C.c_int()
        self.L.orc_pyramid_dims(self.h, level, C.byref(W), C.byref(H))
        buf = np.zeros((H.value, W.value), np.uint8)
        self.L.orc_blurred_copy(self.h, level, _p(buf))
        return buf


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def fast(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size // 4 + 16
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast(_p(img), img.shape[1], img.shape[0], img.strides[0], threshold, _p(out), cap)
    return out[:n].copy()


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(float(y), float(x)))


def gaussian7(img, taps=(18, 34, 48, 56, 48, 34, 18)):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    t = np.asarray(taps, np.uint16)
    lib().orc_gaussian7(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(out), out.strides[0], _p(t))
    return out


def cvt_gray(img, rgb_order):
    img = np.ascontiguousarray(img, np.uint8)
    h, w, c = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_cvt_gray(_p(img), w, h, img.strides[0], c, int(rgb_order), _p(out), w)
    return out


def depth_to_f32(depth_u16, factor):
    d = np.ascontiguousarray(depth_u16, np.uint16)
    out = np.zeros(d.shape, np.float32)
    lib().orc_depth_to_f32(_p(d), d.shape[1], d.shape[0], d.shape[1], C.c_float(factor), _p(out))
    return out


def stereo_from_rgbd(kp, depth_f32, mbf):
    kp = np.ascontiguousarray(kp); d = np.ascontiguousarray(depth_f32, np.float32)
    n = len(kp)
    ur = np.zeros(n, np.float32); dep = np.zeros(n, np.float32)
    lib().orc_stereo_from_rgbd(_p(kp), n, _p(d), d.shape[1], d.shape[0], C.c_float(mbf), _p(ur), _p(dep))
    return ur, dep


def stereo_matches(exL, exR, kpL, descL, kpR, descR, mbf, fx):
    kpL = np.ascontiguousarray(kpL); kpR = np.ascontiguousarray(kpR)
    descL = np.ascontiguousarray(descL, np.uint8); descR = np.ascontiguousarray(descR, np.uint8)
    n = len(kpL)
    ur = np.zeros(n, np.float32); dep = np.zeros(n, np.float32); bd = np.zeros(n, np.int32)
    nm = lib().orc_stereo_matches(exL.h, exR.h, _p(kpL), _p(descL), n, _p(kpR), _p(descR), len(kpR), C.c_float(mbf),
                                  C.c_float(fx), _p(ur), _p(dep), _p(bd))
    return ur, dep, bd, nm


def grid_cells(kp, cam10):
    kp = np.ascontiguousarray(kp); c = np.ascontiguousarray(cam10, np.float32)
    out = np.zeros(len(kp), np.int32)
    lib().orc_grid_cells(_p(kp), len(kp), _p(c), _p(out))
    return out


def unproject(kp, depth, cam10, Twc):
    kp = np.ascontiguousarray(kp); d = np.ascontiguousarray(depth, np.float32)
    c = np.ascontiguousarray(cam10, np.float32); T = np.ascontiguousarray(Twc, np.float32).reshape(16)
    xw = np.zeros((len(kp), 3), np.float32); valid = np.zeros(len(kp), np.uint8)
    lib().orc_unproject(_p(kp), _p(d), len(kp), _p(c), _p(T), _p(xw), _p(valid))
    return xw, valid


def search_by_projection(kpC, descC, uRightC, kpL, dMP, xw, flags, Tcw, Tlw, cam10, scale_factors, th, bMono=False,
                         checkOrientation=True, occupied=None):
    kpC = np.ascontiguousarray(kpC); kpL = np.ascontiguousarray(kpL)
    descC = np.ascontiguousarray(descC, np.uint8); dMP = np.ascontiguousarray(dMP, np.uint8)
    uR = np.ascontiguousarray(uRightC, np.float32); xw = np.ascontiguousarray(xw, np.float32)
    flags = np.ascontiguousarray(flags, np.uint8)
    Tc = np.ascontiguousarray(Tcw, np.float32).reshape(16); Tl = np.ascontiguousarray(Tlw, np.float32).reshape(16)
    c = np.ascontiguousarray(cam10, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
    Nc, Nl = len(kpC), len(kpL)
    match = np.zeros(Nc, np.int32); pairs = np.zeros((Nl, 2), np.int32); npairs = C.c_int()
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    f = lib().orc_search_by_projection
    f.restype = C.c_int
    nm = f(_p(kpC), _p(descC), _p(uR), Nc, _p(kpL), _p(dMP), Nl, _p(xw), _p(flags), _p(Tc), _p(Tl), _p(c), _p(sf),
           C.c_float(th), int(bMono), int(checkOrientation), _p(occ) if occ is not None else None, _p(match), _p(pairs),
           C.byref(npairs))
    return match, pairs[:npairs.value].copy(), nm


MAP_POINT_DTYPE = np.dtype([("xw", "<f4", 3), ("normal", "<f4", 3), ("min_distance", "<f4"), ("max_distance", "<f4"),
                            ("flags", "<u4")])          # sd_map_point, 36 B
TRACK_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("view_cos", "<f4"), ("level", "<i4"),
                        ("in_view", "<i4")])            # sd_track_info, 24 B


def search_local_map(kpF, descF, uRightF, mps, mp_desc, Tcw, cam10, scale_factors, th, nnratio, viewing_cos_limit=0.5,
                     occupied=None):
    """Frame::isInFrustum + ORBmatcher::SearchByProjection(Frame, MapPoints, th) -> (track, mp_match, kp_match, nmatches)."""
    kpF = np.ascontiguousarray(kpF); descF = np.ascontiguousarray(descF, np.uint8)
    uR = np.ascontiguousarray(uRightF, np.float32)
    mps = np.ascontiguousarray(mps, MAP_POINT_DTYPE); md = np.ascontiguousarray(mp_desc, np.uint8)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    c = np.ascontiguousarray(cam10, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
    N, M = len(kpF), len(mps)
    track = np.zeros(M, TRACK_DTYPE); mpm = np.zeros(M, np.int32); kpm = np.zeros(N, np.int32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    f = lib().orc_search_local_map
    f.restype = C.c_int
    nm = f(_p(kpF), _p(descF), _p(uR), N, _p(mps), _p(md), M, _p(T), _p(c), _p(sf), len(sf), C.c_float(th), C.c_float(nnratio),
           C.c_float(viewing_cos_limit), _p(occ) if occ is not None else None, _p(track), _p(mpm), _p(kpm))
    return track, mpm, kpm, nm


def box_track(boxes, last_objects, last_box_idx, last_omit, last_velocity, img_cols, img_rows, cap=None):
    """Frame::boxTrack.  boxes (n,4) f64 -> (boxes', box_idx, omit, velocity) with re-injected boxes appended.  The reference's vectors are
    unbounded: the buffers hold the current boxes plus every last-frame box that could be re-injected."""
    n = len(boxes)
    if cap is None:
        cap = n + len(np.asarray(last_box_idx).reshape(-1)) + 1
    bx = np.zeros((cap, 4), np.float64); bx[:n] = boxes
    lo = np.ascontiguousarray(last_objects, np.float64).reshape(-1, 4)
    li = np.ascontiguousarray(last_box_idx, np.int32); lm = np.ascontiguousarray(last_omit, np.uint8)
    lv = np.ascontiguousarray(last_velocity, np.float64).reshape(-1, 2)
    idx = np.zeros(cap, np.int32); om = np.zeros(cap, np.uint8); vel = np.zeros((cap, 2), np.float64)
    f = lib().orc_box_track
    f.restype = C.c_int
    n2 = f(_p(bx), n, cap, _p(lo), len(lo), _p(li), _p(lm), _p(lv), int(img_cols), int(img_rows), _p(idx), _p(om), _p(vel))
    assert n2 >= 0
    return bx[:n2].copy(), idx[:n2].copy(), om[:n2].copy(), vel[:n2].copy()


def first_separate(kp, desc, boxes, box_idx, omit, velocity):
    """Frame::firstSeparate + ctor split.  Returns dict(kp, desc, perm, Ns, Nd, boxes, box_idx, omit, velocity, boxStart, boxItems)."""
    kp = np.ascontiguousarray(kp).copy(); desc = np.ascontiguousarray(desc, np.uint8).copy()
    N = len(kp); nb = len(boxes)
    bx = np.ascontiguousarray(boxes, np.float64).copy().reshape(-1, 4)
    bi = np.ascontiguousarray(box_idx, np.int32).copy(); bo = np.ascontiguousarray(omit, np.uint8).copy()
    bv = np.ascontiguousarray(velocity, np.float64).copy().reshape(-1, 2)
    perm = np.zeros(N, np.int32); Ns = C.c_int(); Nd = C.c_int()
    boxStart = np.zeros(nb + 1, np.int32); cap = 4 * N + 16
    items = np.zeros(cap, np.int32)
    f = lib().orc_first_separate
    f.restype = C.c_int
    nb2 = f(_p(kp), _p(desc), N, _p(bx), nb, _p(bi), _p(bo), _p(bv), _p(perm), C.byref(Ns), C.byref(Nd), _p(boxStart), _p(items), cap)
    assert nb2 >= 0
    return dict(kp=kp, desc=desc, perm=perm, Ns=Ns.value, Nd=Nd.value, boxes=bx[:nb2], box_idx=bi[:nb2], omit=bo[:nb2],
                velocity=bv[:nb2], boxStart=boxStart[:nb2 + 1].copy(), boxItems=items[:boxStart[nb2]].copy())


def inv3x3(M):
    M = np.ascontiguousarray(M, np.float32).reshape(9); D = np.zeros(9, np.float32)
    lib().orc_inv3x3(_p(M), _p(D))
    return D.reshape(3, 3)


def bf_match_crosscheck(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    m = np.zeros((max(len(q), 1), 2), np.int32)
    f = lib().orc_bf_match_crosscheck
    f.restype = C.c_int
    n = f(_p(q), len(q), _p(t), len(t), _p(m))
    return m[:n].copy()


def separate(HorF, flag, cur, ref, box_idx_last, box_status_last, box_status_cur):
    """Tracking::Separate.  cur/ref: dicts with kp, desc, boxStart, boxItems, box_idx.  Returns
    (ret, box_status_cur', dynStart, dynStatus, matches)."""
    M = np.ascontiguousarray(HorF, np.float32).reshape(9)
    nbC, nbR = len(cur["box_idx"]), len(ref["box_idx"])
    bl = np.ascontiguousarray(box_idx_last, np.int32); sl = np.ascontiguousarray(box_status_last, np.int32)
    sc = np.ascontiguousarray(box_status_cur, np.int32).copy()
    tot = int(cur["boxStart"][-1]) + 1
    dynStart = np.zeros(nbC + 1, np.int32); dyn = np.zeros(tot, np.int32); mt = np.zeros((tot, 2), np.int32)
    f = lib().orc_separate
    f.restype = C.c_int
    args = []
    for fr in (cur, ref):
        args += [np.ascontiguousarray(fr["kp"]), np.ascontiguousarray(fr["desc"], np.uint8),
                 np.ascontiguousarray(fr["boxStart"], np.int32), np.ascontiguousarray(fr["boxItems"], np.int32),
                 np.ascontiguousarray(fr["box_idx"], np.int32)]
    ret = f(_p(M), int(flag), _p(args[0]), _p(args[1]), _p(args[2]), _p(args[3]), _p(args[4]), nbC,
            _p(args[5]), _p(args[6]), _p(args[7]), _p(args[8]), _p(args[9]), nbR, _p(bl), _p(sl), len(bl), _p(sc),
            _p(dynStart), _p(dyn), _p(mt))
    n = dynStart[nbC]
    return ret, sc, dynStart, dyn[:n].copy(), mt[:n].copy()


def update_frame(kp, boxStart, boxItems, dynStart, dynStatus):
    kp = np.ascontiguousarray(kp); bs = np.ascontiguousarray(boxStart, np.int32); it = np.ascontiguousarray(boxItems, np.int32)
    ds = np.ascontiguousarray(dynStart, np.int32); dy = np.ascontiguousarray(dynStatus, np.int32)
    out = np.zeros(len(dy) + 1, np.int32)
    f = lib().orc_update_frame
    f.restype = C.c_int
    n = f(_p(kp), _p(bs), _p(it), len(bs) - 1, _p(ds), _p(dy), _p(out))
    return out[:n].copy()


# ---- bag of words (oracle/bow_oracle.inc) ----
class Vocabulary:
    def __init__(self, handle):
        if not handle:
            raise ValueError("vocabulary could not be built / loaded")
        self.h = C.c_void_p(handle)

    @classmethod
    def from_nodes(cls, voc):
        f = lib().orc_vocab_create
        f.restype = C.c_void_p
        par = np.ascontiguousarray(voc["parent"], np.int32); leaf = np.ascontiguousarray(voc["is_leaf"], np.uint8)
        d = np.ascontiguousarray(voc["desc"], np.uint8); w = np.ascontiguousarray(voc["weight"], np.float64)
        return cls(f(int(voc["k"]), int(voc["L"]), int(voc["scoring"]), int(voc["weighting"]), len(par), _p(par), _p(leaf), _p(d), _p(w)))

    @classmethod
    def load_text(cls, path):
        f = lib().orc_vocab_load_text
        f.restype = C.c_void_p
        return cls(f(str(path).encode()))

    def info(self):
        h = np.zeros(4, np.int32)
        lib().orc_vocab_header(self.h, _p(h))
        return dict(k=int(h[0]), L=int(h[1]), scoring=int(h[2]), weighting=int(h[3]), n_nodes=int(lib().orc_vocab_nodes(self.h)),
                    n_words=int(lib().orc_vocab_words(self.h)))

    def nodes(self):
        n = self.info()["n_nodes"]
        par = np.zeros(n, np.int32); nch = np.zeros(n, np.int32); wid = np.zeros(n, np.int32)
        d = np.zeros((n, 32), np.uint8); w = np.zeros(n, np.float64)
        lib().orc_vocab_export(self.h, _p(par), _p(nch), _p(wid), _p(d), _p(w))
        return dict(parent=par, n_children=nch, word_id=wid, desc=d, weight=w)

    def transform(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8); N = len(desc)
        word = np.zeros(N, np.uint32); weight = np.zeros(N, np.float64); nid = np.zeros(N, np.uint32)
        lib().orc_transform(self.h, _p(desc), N, levelsup, _p(word), _p(weight), _p(nid))
        return word, weight, nid

    def compute_bow(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8); N = len(desc)
        bw = np.zeros(N, np.uint32); bv = np.zeros(N, np.float64); fn = np.zeros(N, np.uint32); ff = np.zeros(N, np.uint32)
        nf = C.c_int()
        nb = lib().orc_compute_bow(self.h, _p(desc), N, levelsup, _p(bw), _p(bv), _p(fn), _p(ff), C.byref(nf))
        return dict(word=bw[:nb].copy(), value=bv[:nb].copy(), fv_node=fn[:nf.value].copy(), fv_feature=ff[:nf.value].copy())

    def close(self):
        if self.h:
            lib().orc_vocab_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bow_score_l1(a, b):
    f = lib().orc_bow_score_l1
    f.restype = C.c_double
    w1 = np.ascontiguousarray(a["word"], np.uint32); v1 = np.ascontiguousarray(a["value"], np.float64)
    w2 = np.ascontiguousarray(b["word"], np.uint32); v2 = np.ascontiguousarray(b["value"], np.float64)
    return float(f(_p(w1), _p(v1), len(w1), _p(w2), _p(v2), len(w2)))


def search_by_bow(kpKF, descKF, kf_valid, bowKF, kpF, descF, bowF, nnratio, checkOrientation=True):
    kpKF = np.ascontiguousarray(kpKF); kpF = np.ascontiguousarray(kpF)
    dK = np.ascontiguousarray(descKF, np.uint8); dF = np.ascontiguousarray(descF, np.uint8)
    val = np.ascontiguousarray(kf_valid, np.uint8)
    nK = np.ascontiguousarray(bowKF["fv_node"], np.uint32); fK = np.ascontiguousarray(bowKF["fv_feature"], np.uint32)
    nF = np.ascontiguousarray(bowF["fv_node"], np.uint32); fF = np.ascontiguousarray(bowF["fv_feature"], np.uint32)
    match = np.zeros(len(kpF), np.int32)
    f = lib().orc_search_by_bow
    f.restype = C.c_int
    nm = f(_p(kpKF), _p(dK), _p(val), _p(nK), _p(fK), len(nK), _p(kpF), _p(dF), len(kpF), _p(nF), _p(fF), len(nF), C.c_float(nnratio),
           int(checkOrientation), _p(match))
    return match, nm


def estimate_motion(points_last, points_current):
    """The H / F fit of Tracking::TrackHomo (spec Q13) -> dict(H, F, mask_h, mask_f, n_h, n_f, HorF, flag)."""
    p1 = np.ascontiguousarray(points_last, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(points_current, np.float32).reshape(-1, 2)
    N = len(p1)
    H = np.zeros(9, np.float64); F = np.zeros(9, np.float64); mh = np.zeros(max(N, 1), np.uint8); mf = np.zeros(max(N, 1), np.uint8)
    nh, nf = C.c_int(), C.c_int(); hf = np.zeros(9, np.float32)
    f = lib().orc_estimate_motion
    f.restype = C.c_int
    flag = f(_p(p1), _p(p2), N, _p(H), _p(F), _p(mh), _p(mf), C.byref(nh), C.byref(nf), _p(hf))
    return dict(H=H.reshape(3, 3), F=F.reshape(3, 3), mask_h=mh[:N], mask_f=mf[:N], n_h=nh.value, n_f=nf.value, HorF=hf.reshape(3, 3), flag=flag)


def undistort_points(pts, K4, dist5):
    """cv::undistortPoints(pts, mK, mDistCoef, Mat(), mK) of Frame::UndistortKeyPoints -> (N, 2) f32."""
    p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2); out = np.zeros_like(p)
    k = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist5, np.float32)
    lib().orc_undistort_points(_p(p), len(p), _p(k), _p(d), _p(out))
    return out


def image_bounds(cols, rows, K4, dist5):
    k = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist5, np.float32); b = np.zeros(4, np.float32)
    lib().orc_image_bounds(int(cols), int(rows), _p(k), _p(d), _p(b))
    return b
