"""ctypes binding of libsd_frontend.so (the C ABI in include/sd_frontend.h).

This is plumbing for tests and bench.py: the product is the shared library.  Class and
method names follow the reference's C++ API (ORB_SLAM2::ORBextractor, Frame, ORBmatcher) so
that the parity tests read like calls into the reference.  There is NO CPU fallback: if the
library is missing, or no HIP device is present, calls raise.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SD_FRONTEND_LIB") or os.path.join(_HERE, "lib", "libsd_frontend.so")     # the override is for kernel experiments only

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])

SD_OK = 0
SD_ERR_INVALID, SD_ERR_NO_DEVICE, SD_ERR_HIP, SD_ERR_CAPACITY, SD_ERR_UNSUPPORTED, SD_ERR_STATE = -1, -2, -3, -4, -5, -6


class SdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sd_frontend error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load the HIP extension; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                              % LIB_PATH)
        try:
            # torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's): import it first so the
            # process holds ONE HIP runtime shared by torch (device memory, RCCL) and this library.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.sd_status_string.restype = C.c_char_p
        L.sd_last_error.restype = C.c_char_p
        vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        L.sd_extractor_create.argtypes = [C.POINTER(vp), i, f, i, i, i]
        L.sd_extractor_destroy.argtypes = [vp]
        L.sd_extractor_set_blur_taps.argtypes = [vp, vp]
        L.sd_extractor_levels.argtypes = [vp, C.POINTER(i), C.POINTER(f)]
        L.sd_extractor_tables.argtypes = [vp] * 7
        L.sd_extractor_level_size.argtypes = [vp, i, i, i, C.POINTER(i), C.POINTER(i)]
        L.sd_batch_create.argtypes = [C.POINTER(vp), vp, i, i, i]
        L.sd_batch_destroy.argtypes = [vp]
        L.sd_batch_kp_capacity.argtypes = [vp, C.POINTER(i)]
        L.sd_batch_extract_device.argtypes = [vp, vp, sz, sz, i, vp]
        L.sd_batch_extract_color_device.argtypes = [vp, vp, sz, sz, i, i, vp]
        L.sd_batch_extract_pixels_device.argtypes = [vp, vp, sz, sz, i, i, i, vp]
        L.sd_batch_rgbd_from_f32_scaled.argtypes = [vp, vp, sz, sz, i, f, f, vp]
        L.sd_tracker_set_mappoints.argtypes = [vp, vp, vp, vp]
        L.sd_tracker_set_state.argtypes = [vp, vp]
        L.sd_tracker_prefetch.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, vp]
        L.sd_tracker_prefetched_record_bytes.argtypes = [vp]
        L.sd_tracker_prefetched_record_bytes.restype = sz
        L.sd_tracker_export_prefetched.argtypes = [vp, i, i, vp, sz, vp]
        L.sd_tracker_import_prefetched.argtypes = [vp, vp, sz, i, vp]
        L.sd_tracker_discard_prefetched.argtypes = [vp]
        L.sd_batch_extract_host.argtypes = [vp, vp, sz, sz, i]
        L.sd_batch_results_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i)]
        L.sd_batch_counts.argtypes = [vp, vp, i]
        L.sd_batch_download.argtypes = [vp, i, vp, vp, i, C.POINTER(i), vp]
        L.sd_batch_pyramid_level.argtypes = [vp, i, i, C.POINTER(vp), C.POINTER(i), C.POINTER(i), C.POINTER(sz)]
        L.sd_batch_download_pyramid.argtypes = [vp, i, i, vp]
        L.sd_batch_download_blurred.argtypes = [vp, i, i, vp]
        L.sd_batch_candidate_counts.argtypes = [vp, i, vp]
        L.sd_batch_stereo_match.argtypes = [vp, i, f, f, vp]
        L.sd_batch_stereo_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i)]
        L.sd_batch_download_stereo.argtypes = [vp, i, vp, vp, vp, i]
        L.sd_batch_rgbd_from_u16.argtypes = [vp, vp, sz, sz, i, f, f, vp]
        L.sd_batch_rgbd_from_f32.argtypes = [vp, vp, sz, sz, i, f, vp]
        L.sd_batch_download_rgbd.argtypes = [vp, i, vp, vp, i]
        L.sd_batch_assign_grid.argtypes = [vp, i, vp, vp]
        L.sd_batch_download_grid.argtypes = [vp, i, vp, i]
        L.sd_batch_unproject.argtypes = [vp, i, i, i, vp, vp, vp]
        L.sd_batch_mappoints_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i)]
        L.sd_batch_set_mappoints.argtypes = [vp, i, vp, vp, i]
        L.sd_batch_download_mappoints.argtypes = [vp, i, vp, vp, i]
        L.sd_batch_search_by_projection.argtypes = [vp, i, vp, vp, vp, vp, vp, f, i, i, vp, vp, vp]
        L.sd_batch_copy_frame.argtypes = [vp, i, i, vp]
        L.sd_batch_matches_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i)]
        L.sd_batch_download_matches.argtypes = [vp, i, vp, vp, i, C.POINTER(i), C.POINTER(i)]
        L.sd_box_track.argtypes = [vp, i, i, vp, i, vp, vp, vp, i, i, vp, vp, vp, C.POINTER(i)]
        L.sd_batch_first_separate.argtypes = [vp, i, vp, vp, vp, vp, vp]
        L.sd_batch_download_boxes.argtypes = [vp, i, C.POINTER(i), vp, vp, vp, vp, vp, vp, i, C.POINTER(i), C.POINTER(i)]
        L.sd_batch_download_dynamic.argtypes = [vp, i, vp, vp, vp, vp, i, C.POINTER(i)]
        L.sd_batch_separate.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, vp, vp]
        L.sd_batch_download_separate.argtypes = [vp, i, C.POINTER(i), vp, vp, vp, i]
        L.sd_batch_update_frame.argtypes = [vp, i, vp]
        L.sd_refqueue_create.argtypes = [C.POINTER(vp)]
        L.sd_refqueue_destroy.argtypes = [vp]
        L.sd_refqueue_clear.argtypes = [vp]
        L.sd_refqueue_size.argtypes = [vp, C.POINTER(i)]
        L.sd_refqueue_candidate.argtypes = [vp, C.c_double, i, C.POINTER(i)]
        L.sd_refqueue_reject.argtypes = [vp, C.POINTER(i)]
        L.sd_refqueue_push.argtypes = [vp, C.c_double, i, i, i, C.POINTER(i)]
        L.sd_cvt_gray_device.argtypes = [vp, i, i, sz, sz, i, i, vp, sz, sz, i, vp]
        L.sd_depth_to_f32_device.argtypes = [vp, i, i, sz, f, vp, i, sz, vp]
        L.sd_descriptor_distance.argtypes = [vp, vp]
        L.sd_hamming_matrix_device.argtypes = [vp, i, vp, i, vp, vp]
        L.sd_batch_set_profiling.argtypes = [vp, i]
        L.sd_batch_kernel_count.argtypes = [vp, C.POINTER(i)]
        L.sd_batch_kernel_times.argtypes = [vp, i, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.sd_batch_reset_kernel_times.argtypes = [vp]
        L.sd_batch_sync.argtypes = [vp]
        L.sd_device_count.argtypes = [C.POINTER(i)]
        L.sd_tracker_create.argtypes = [C.POINTER(vp), vp, vp]
        L.sd_tracker_destroy.argtypes = [vp]
        L.sd_tracker_reset.argtypes = [vp]
        L.sd_tracker_batch.argtypes = [vp, C.POINTER(vp)]
        L.sd_tracker_track.argtypes = [vp, vp, sz, sz, vp, sz, sz, vp, vp, vp, vp, vp, vp, vp]
        L.sd_batch_copy_frames.argtypes = [vp, i, vp, vp, vp]
        L.sd_batch_boxes_device.argtypes = [vp, C.POINTER(vp)]
        L.sd_batch_set_distortion.argtypes = [vp, vp, vp]
        L.sd_batch_undistort.argtypes = [vp, i, vp, vp]
        L.sd_batch_download_keys_un.argtypes = [vp, i, vp, i, C.POINTER(i)]
        L.sd_image_bounds.argtypes = [i, i, vp, vp, vp]
        L.sd_batch_backproject_dense.argtypes = [vp, i, vp, vp, sz, sz, vp, sz, sz, f, vp, sz, sz, vp, vp, vp, i, vp, vp]
        _lib = L
    return _lib


def check(rc):
    if rc != SD_OK:
        L = lib()
        raise SdError(rc, (L.sd_status_string(rc) or b"").decode() + ": " + (L.sd_last_error() or b"").decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ORBextractor:
    """ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-114): parameters and derived tables."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST):
        L = lib()
        self.h = C.c_void_p()
        check(L.sd_extractor_create(C.byref(self.h), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST))
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.mvScaleFactor = np.zeros(nlevels, np.float32); self.mvInvScaleFactor = np.zeros(nlevels, np.float32)
        self.mvLevelSigma2 = np.zeros(nlevels, np.float32); self.mvInvLevelSigma2 = np.zeros(nlevels, np.float32)
        self.mnFeaturesPerLevel = np.zeros(nlevels, np.int32); self.umax = np.zeros(16, np.int32)
        check(L.sd_extractor_tables(self.h, _p(self.mvScaleFactor), _p(self.mvInvScaleFactor), _p(self.mvLevelSigma2),
                                    _p(self.mvInvLevelSigma2), _p(self.mnFeaturesPerLevel), _p(self.umax)))

    def __del__(self):
        try:
            lib().sd_extractor_destroy(self.h)
        except Exception:
            pass

    def GetLevels(self):
        n = C.c_int(); s = C.c_float()
        check(lib().sd_extractor_levels(self.h, C.byref(n), C.byref(s)))
        return n.value

    def GetScaleFactor(self):
        n = C.c_int(); s = C.c_float()
        check(lib().sd_extractor_levels(self.h, C.byref(n), C.byref(s)))
        return s.value

    def set_blur_taps(self, taps):
        t = np.asarray(taps, np.uint16)
        check(lib().sd_extractor_set_blur_taps(self.h, _p(t)))

    def level_size(self, width, height, level):
        w = C.c_int(); h = C.c_int()
        check(lib().sd_extractor_level_size(self.h, width, height, level, C.byref(w), C.byref(h)))
        return w.value, h.value


class Batch:
    """Device workspace: batched ORBextractor::operator() + the Frame-side association steps."""

    def __init__(self, extractor, width, height, max_images, handle=None):
        self.ex = extractor
        self.W, self.H, self.max_images = width, height, max_images
        self.owned = handle is None
        self.h = C.c_void_p()
        if handle is None:
            check(lib().sd_batch_create(C.byref(self.h), extractor.h, width, height, max_images))
        else:                                  # a view of the workspace owned by a Tracker
            self.h = C.c_void_p(handle)
        c = C.c_int()
        check(lib().sd_batch_kp_capacity(self.h, C.byref(c)))
        self.cap = c.value

    def close(self):
        if self.h and self.owned:
            lib().sd_batch_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- extraction
    def extract_host(self, images):
        """images: (n, H, W) u8 array (or list of (H, W) arrays)."""
        arr = np.ascontiguousarray(np.stack(images) if isinstance(images, (list, tuple)) else images, np.uint8)
        if arr.ndim == 2:
            arr = arr[None]
        n, h, w = arr.shape
        assert (h, w) == (self.H, self.W)
        check(lib().sd_batch_extract_host(self.h, _p(arr), w, w * h, n))
        return n

    def extract_device(self, d_ptr, stride, pitch, n, stream=None):
        check(lib().sd_batch_extract_device(self.h, C.c_void_p(d_ptr), stride, pitch, n, C.c_void_p(stream or 0)))

    def extract_color_device(self, d_ptr, stride, pitch, n, rgb_order=True, stream=None):
        """cvtColor + operator() for 3-channel images in HBM (GrabImageRGBD's conversion fused into pyramid level 0)."""
        check(lib().sd_batch_extract_color_device(self.h, C.c_void_p(d_ptr), stride, pitch, int(bool(rgb_order)), n,
                                                  C.c_void_p(stream or 0)))

    def extract_pixels_device(self, d_ptr, stride, pitch, n, channels, rgb_order=True, stream=None):
        """Any input GrabImage* accepts: 1 (gray), 3 or 4 channels (Tracking.cc:175-200)."""
        check(lib().sd_batch_extract_pixels_device(self.h, C.c_void_p(d_ptr), stride, pitch, channels, int(bool(rgb_order)), n,
                                                   C.c_void_p(stream or 0)))

    def sync(self):
        check(lib().sd_batch_sync(self.h))

    def counts(self, n):
        out = np.zeros(n, np.int32)
        check(lib().sd_batch_counts(self.h, _p(out), n))
        return out

    def download(self, image):
        kp = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8)
        per_level = np.zeros(self.ex.nlevels, np.int32)
        n = C.c_int()
        check(lib().sd_batch_download(self.h, image, _p(kp), _p(desc), self.cap, C.byref(n), _p(per_level)))
        return kp[:n.value].copy(), desc[:n.value].copy(), per_level

    def download_keys_un(self, image):
        """mvKeysUn of a slot (== mvKeys unless sd_batch_set_distortion was called with Camera.k1 != 0)."""
        kp = np.zeros(self.cap, KP_DTYPE)
        n = C.c_int()
        check(lib().sd_batch_download_keys_un(self.h, image, _p(kp), self.cap, C.byref(n)))
        return kp[:n.value].copy()

    def set_distortion(self, K4, dist5):
        k = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist5, np.float32)
        check(lib().sd_batch_set_distortion(self.h, _p(k), _p(d)))

    def undistort(self, slots, stream=None):
        sl = np.ascontiguousarray(slots, np.int32)
        check(lib().sd_batch_undistort(self.h, len(sl), _p(sl), C.c_void_p(stream or 0)))

    def pyramid(self, image, level):
        w, h = self.ex.level_size(self.W, self.H, level)
        out = np.zeros((h + 38, w + 38), np.uint8)
        check(lib().sd_batch_download_pyramid(self.h, image, level, _p(out)))
        return out

    def blurred(self, image, level):
        w, h = self.ex.level_size(self.W, self.H, level)
        out = np.zeros((h, w), np.uint8)
        check(lib().sd_batch_download_blurred(self.h, image, level, _p(out)))
        return out

    def candidate_counts(self, image):
        out = np.zeros(self.ex.nlevels, np.int32)
        check(lib().sd_batch_candidate_counts(self.h, image, _p(out)))
        return out

    def results_device(self):
        kp, desc, cnt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap = C.c_int()
        check(lib().sd_batch_results_device(self.h, C.byref(kp), C.byref(desc), C.byref(cnt), C.byref(cap)))
        return kp.value, desc.value, cnt.value, cap.value

    # -- Frame::ComputeStereoMatches
    def stereo_match(self, n_frames, mbf, fx, stream=None):
        check(lib().sd_batch_stereo_match(self.h, n_frames, mbf, fx, C.c_void_p(stream or 0)))

    def download_stereo(self, frame):
        ur = np.zeros(self.cap, np.float32); dep = np.zeros(self.cap, np.float32); sad = np.zeros(self.cap, np.int32)
        check(lib().sd_batch_download_stereo(self.h, frame, _p(ur), _p(dep), _p(sad), self.cap))
        return ur, dep, sad

    # -- Frame::ComputeStereoFromRGBD
    def rgbd_from_u16(self, d_depth_ptr, stride_elems, pitch_elems, n, depth_factor, mbf, stream=None):
        check(lib().sd_batch_rgbd_from_u16(self.h, C.c_void_p(d_depth_ptr), stride_elems, pitch_elems, n, depth_factor,
                                           mbf, C.c_void_p(stream or 0)))

    def rgbd_from_f32(self, d_depth_ptr, stride_elems, pitch_elems, n, mbf, stream=None):
        check(lib().sd_batch_rgbd_from_f32(self.h, C.c_void_p(d_depth_ptr), stride_elems, pitch_elems, n, mbf,
                                           C.c_void_p(stream or 0)))

    def download_rgbd(self, image):
        ur = np.zeros(self.cap, np.float32); dep = np.zeros(self.cap, np.float32)
        check(lib().sd_batch_download_rgbd(self.h, image, _p(ur), _p(dep), self.cap))
        return ur, dep

    # -- Frame grid / UnprojectStereo / ORBmatcher::SearchByProjection(Frame, Frame)
    def assign_grid(self, n_images, cam, stream=None):
        c = camera_array(cam)
        check(lib().sd_batch_assign_grid(self.h, n_images, _p(c), C.c_void_p(stream or 0)))

    def download_grid(self, image):
        out = np.zeros(self.cap, np.int16)
        check(lib().sd_batch_download_grid(self.h, image, _p(out), self.cap))
        return out

    def unproject(self, image_step, n_frames, cam, Twc, stream=None):
        c = camera_array(cam)
        T = np.ascontiguousarray(Twc, np.float32).reshape(n_frames, 16)
        check(lib().sd_batch_unproject(self.h, 0, image_step, n_frames, _p(c), _p(T), C.c_void_p(stream or 0)))

    def set_mappoints(self, image, xw, flags):
        xw = np.ascontiguousarray(xw, np.float32); flags = np.ascontiguousarray(flags, np.uint8)
        check(lib().sd_batch_set_mappoints(self.h, image, _p(xw), _p(flags), len(flags)))

    def download_mappoints(self, image):
        xw = np.zeros((self.cap, 3), np.float32); fl = np.zeros(self.cap, np.uint8)
        check(lib().sd_batch_download_mappoints(self.h, image, _p(xw), _p(fl), self.cap))
        return xw, fl

    def search_by_projection(self, cur_index, last_index, Tcw, Tlw, cam, th, bMono=False,
                             checkOrientation=True, d_occupied=None, d_mp_desc=None, stream=None):
        c = camera_array(cam)
        ci = np.ascontiguousarray(cur_index, np.int32); li = np.ascontiguousarray(last_index, np.int32)
        n_pairs = len(ci)
        Tc = np.ascontiguousarray(Tcw, np.float32).reshape(n_pairs, 16)
        Tl = np.ascontiguousarray(Tlw, np.float32).reshape(n_pairs, 16)
        check(lib().sd_batch_search_by_projection(self.h, n_pairs, _p(ci), _p(li), _p(Tc), _p(Tl), _p(c),
                                                  th, int(bMono), int(checkOrientation), C.c_void_p(d_occupied or 0),
                                                  C.c_void_p(d_mp_desc or 0), C.c_void_p(stream or 0)))

    def search_local_map(self, frame_index, point_offset, d_points, d_point_desc, Tcw, cam, th, nnratio, d_track,
                         d_point_match, d_kp_match, d_nmatches, viewing_cos_limit=0.5, d_occupied=None, stream=None):
        """Tracking::SearchLocalPoints: Frame::isInFrustum + ORBmatcher::SearchByProjection(Frame, MapPoints, th).
        d_* are device pointers (ints); see include/sd_frontend.h for the layouts."""
        c = camera_array(cam)
        fi = np.ascontiguousarray(frame_index, np.int32); po = np.ascontiguousarray(point_offset, np.int32)
        n = len(fi)
        T = np.ascontiguousarray(Tcw, np.float32).reshape(n, 16)
        check(lib().sd_batch_search_local_map(self.h, n, _p(fi), _p(po), C.c_void_p(d_points), C.c_void_p(d_point_desc), _p(T),
                                              _p(c), C.c_float(th), C.c_float(nnratio), C.c_float(viewing_cos_limit),
                                              C.c_void_p(d_occupied or 0), C.c_void_p(d_track), C.c_void_p(d_point_match),
                                              C.c_void_p(d_kp_match), C.c_void_p(d_nmatches), C.c_void_p(stream or 0)))

    # -- Frame::ComputeBoW, ORBmatcher::SearchByBoW
    def compute_bow(self, vocab, image_index, levelsup=4, stream=None):
        ii = np.ascontiguousarray(image_index, np.int32)
        check(lib().sd_batch_compute_bow(self.h, vocab.h, len(ii), _p(ii), levelsup, C.c_void_p(stream or 0)))

    def download_bow(self, image):
        """-> dict(word, value: the BowVector; fv_node, fv_feature: the flattened FeatureVector; f_word, f_weight, f_node: per feature)"""
        cap = self.cap
        bw = np.zeros(cap, np.uint32); bv = np.zeros(cap, np.float64); fn = np.zeros(cap, np.uint32); ff = np.zeros(cap, np.uint32)
        fw = np.zeros(cap, np.uint32); fwt = np.zeros(cap, np.float64); fnd = np.zeros(cap, np.uint32)
        nw, nf = C.c_int(), C.c_int()
        check(lib().sd_batch_download_bow(self.h, image, _p(bw), _p(bv), C.byref(nw), _p(fn), _p(ff), C.byref(nf), _p(fw), _p(fwt), _p(fnd), cap))
        n = int(self.counts(image + 1)[image])
        return dict(word=bw[:nw.value].copy(), value=bv[:nw.value].copy(), fv_node=fn[:nf.value].copy(), fv_feature=ff[:nf.value].copy(),
                    f_word=fw[:n].copy(), f_weight=fwt[:n].copy(), f_node=fnd[:n].copy())

    def search_by_bow(self, kf_index, frame_index, nnratio, checkOrientation=True, d_kf_valid=None, stream=None):
        ki = np.ascontiguousarray(kf_index, np.int32); fi = np.ascontiguousarray(frame_index, np.int32)
        check(lib().sd_batch_search_by_bow(self.h, len(ki), _p(ki), _p(fi), C.c_void_p(d_kf_valid or 0), C.c_float(nnratio),
                                           int(checkOrientation), C.c_void_p(stream or 0)))

    # -- Tracking::TrackHomo model fit (H / F from the projection matcher's point pairs)
    def estimate_motion(self, stream=None):
        check(lib().sd_batch_estimate_motion(self.h, C.c_void_p(stream or 0)))

    def download_motion(self, pair):
        H = np.zeros(9, np.float64); F = np.zeros(9, np.float64); mh = np.zeros(self.cap, np.uint8); mf = np.zeros(self.cap, np.uint8)
        n, nh, nf, flag = C.c_int(), C.c_int(), C.c_int(), C.c_int(); hf = np.zeros(9, np.float32)
        check(lib().sd_batch_download_motion(self.h, pair, _p(H), _p(F), _p(mh), _p(mf), self.cap, C.byref(n), C.byref(nh), C.byref(nf),
                                             _p(hf), C.byref(flag)))
        return dict(H=H.reshape(3, 3), F=F.reshape(3, 3), mask_h=mh[:n.value].copy(), mask_f=mf[:n.value].copy(), n_h=nh.value, n_f=nf.value,
                    HorF=hf.reshape(3, 3), flag=flag.value)

    def copy_frame(self, src, dst, stream=None):
        check(lib().sd_batch_copy_frame(self.h, src, dst, C.c_void_p(stream or 0)))

    def copy_frames(self, src, dst, stream=None):
        a = np.ascontiguousarray(src, np.int32); d = np.ascontiguousarray(dst, np.int32)
        check(lib().sd_batch_copy_frames(self.h, len(a), _p(a), _p(d), C.c_void_p(stream or 0)))

    def download_matches(self, pair):
        match = np.zeros(self.cap, np.int32); pairs = np.zeros((self.cap, 2), np.int32)
        npairs = C.c_int(); nm = C.c_int()
        check(lib().sd_batch_download_matches(self.h, pair, _p(match), _p(pairs), self.cap, C.byref(npairs), C.byref(nm)))
        return match, pairs[:npairs.value].copy(), nm.value

    # -- dynamic-object cull: Frame::firstSeparate, Tracking::Separate, Frame::UpdateFrame
    @staticmethod
    def pack_boxes(boxes_list, box_idx_list):
        n = len(boxes_list)
        bx = np.zeros((n, MAXB, 4), np.float64); bi = np.zeros((n, MAXB), np.int32); nb = np.zeros(n, np.int32)
        for k in range(n):
            m = len(boxes_list[k]); nb[k] = m
            if m:
                bx[k, :m] = np.asarray(boxes_list[k], np.float64).reshape(m, 4); bi[k, :m] = box_idx_list[k]
        return bx, nb, bi

    def first_separate(self, slots, boxes_list, box_idx_list, stream=None, packed=None):
        sl = np.ascontiguousarray(slots, np.int32)
        bx, nb, bi = packed if packed is not None else self.pack_boxes(boxes_list, box_idx_list)
        check(lib().sd_batch_first_separate(self.h, len(sl), _p(sl), _p(bx), _p(nb), _p(bi), C.c_void_p(stream or 0)))

    def download_boxes(self, slot):
        nb = C.c_int(); n_all = C.c_int(); n_s = C.c_int()
        bx = np.zeros((MAXB, 4), np.float64); bi = np.zeros(MAXB, np.int32); bs = np.zeros(MAXB, np.int32)
        ko = np.zeros(MAXB, np.int32); st = np.zeros(MAXB + 1, np.int32); it = np.zeros(2 * self.cap, np.int32)
        check(lib().sd_batch_download_boxes(self.h, slot, C.byref(nb), _p(bx), _p(bi), _p(bs), _p(ko), _p(st), _p(it), len(it),
                                            C.byref(n_all), C.byref(n_s)))
        m = nb.value
        return dict(nb=m, boxes=bx[:m].copy(), box_idx=bi[:m].copy(), box_status=bs[:m].copy(), kept_orig=ko[:m].copy(),
                    boxStart=st[:m + 1].copy(), boxItems=it[:st[m]].copy(), n_all=n_all.value, n_static=n_s.value)

    def download_dynamic(self, slot):
        kp = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8)
        ur = np.zeros(self.cap, np.float32); dep = np.zeros(self.cap, np.float32)
        n = C.c_int()
        check(lib().sd_batch_download_dynamic(self.h, slot, _p(kp), _p(desc), _p(ur), _p(dep), self.cap, C.byref(n)))
        m = n.value
        return kp[:m].copy(), desc[:m].copy(), ur[:m].copy(), dep[:m].copy()

    def separate(self, cur_index, ref_index, HorF, flag, last_box_idx, last_box_status, stream=None, packed_last=None):
        n = len(cur_index)
        ci = np.ascontiguousarray(cur_index, np.int32); ri = np.ascontiguousarray(ref_index, np.int32)
        if HorF is None:                     # HorF / flag from the preceding estimate_motion, on the device
            M = fl = None
        else:
            M = np.ascontiguousarray(HorF, np.float32).reshape(n, 9); fl = np.ascontiguousarray(flag, np.int32)
        if packed_last is not None:
            li, ls, nl = packed_last
        else:
            li = np.zeros((n, MAXB), np.int32); ls = np.zeros((n, MAXB), np.int32); nl = np.zeros(n, np.int32)
            for k in range(n):
                m = len(last_box_idx[k]); nl[k] = m
                li[k, :m] = last_box_idx[k]; ls[k, :m] = last_box_status[k]
        check(lib().sd_batch_separate(self.h, n, _p(ci), _p(ri), _p(M) if M is not None else None, _p(fl) if fl is not None else None, _p(li), _p(ls),
                                      _p(nl), C.c_void_p(stream or 0)))

    def download_separate(self, pair):
        ret = C.c_int(); ds = np.zeros(MAXB + 1, np.int32)
        dyn = np.zeros(2 * self.cap, np.int32); mt = np.zeros((2 * self.cap, 2), np.int32)
        check(lib().sd_batch_download_separate(self.h, pair, C.byref(ret), _p(ds), _p(dyn), _p(mt), len(dyn)))
        n = ds[MAXB]
        return ret.value, ds, dyn[:n].copy(), mt[:n].copy()

    def update_frame(self, only_if_static=True, stream=None):
        check(lib().sd_batch_update_frame(self.h, int(only_if_static), C.c_void_p(stream or 0)))

    # -- PointCloudMapping::generatePointCloud
    def backproject_dense(self, slots, d_color, color_stride, color_pitch, d_depth, depth_stride, depth_pitch, depth_factor, d_mask, mask_stride,
                          mask_pitch, cam, Twc, d_points, cap_points, d_counts, stream=None):
        sl = np.ascontiguousarray(slots, np.int32); c = camera_array(cam)
        T = np.ascontiguousarray(Twc, np.float64).reshape(len(sl), 16)
        check(lib().sd_batch_backproject_dense(self.h, len(sl), _p(sl), C.c_void_p(d_color), color_stride, color_pitch, C.c_void_p(d_depth), depth_stride,
                                               depth_pitch, C.c_float(depth_factor), C.c_void_p(d_mask or 0), mask_stride, mask_pitch, _p(c), _p(T),
                                               C.c_void_p(d_points), cap_points, C.c_void_p(d_counts), C.c_void_p(stream or 0)))

    # -- profiling
    def set_profiling(self, on):
        check(lib().sd_batch_set_profiling(self.h, int(bool(on))))

    def reset_kernel_times(self):
        check(lib().sd_batch_reset_kernel_times(self.h))

    def kernel_times(self):
        n = C.c_int()
        check(lib().sd_batch_kernel_count(self.h, C.byref(n)))
        out = {}
        for i in range(n.value):
            name = C.c_char_p(); ms = C.c_double(); cnt = C.c_int64()
            check(lib().sd_batch_kernel_times(self.h, i, C.byref(name), C.byref(ms), C.byref(cnt)))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out


MAXB = 64             # SD_MAX_BOXES (include/sd_frontend.h)
CLOUD_POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1")])     # sd_cloud_point
SENSOR_MONOCULAR, SENSOR_STEREO, SENSOR_RGBD = 0, 1, 2


class _Camera(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "mbf", "mb", "mnMinX", "mnMaxX", "mnMinY", "mnMaxY")]


class TrackerParams(C.Structure):
    """sd_tracker_params (include/sd_frontend.h)."""
    _fields_ = [("sensor", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("rgb_order", C.c_int32),
                ("n_lanes", C.c_int32), ("track_last", C.c_int32), ("depth_type", C.c_int32), ("cam", _Camera), ("dist", C.c_float * 5),
                ("fps", C.c_float), ("depth_map_factor", C.c_float), ("th_depth", C.c_float), ("ini_features", C.c_int32), ("lookahead", C.c_int32)]


class LaneResult(C.Structure):
    """sd_lane_result (include/sd_frontend.h)."""
    _fields_ = [("frame_id", C.c_int32), ("cur_slot", C.c_int32), ("last_slot", C.c_int32), ("ref_slot", C.c_int32), ("ref_frame_id", C.c_int32),
                ("track_flag", C.c_int32), ("separate_ret", C.c_int32), ("n_track_matches", C.c_int32), ("n_track_pairs", C.c_int32),
                ("n_h", C.c_int32), ("n_f", C.c_int32), ("n_last_matches", C.c_int32), ("N", C.c_int32), ("N_s", C.c_int32), ("N_d", C.c_int32),
                ("n_boxes", C.c_int32), ("box_idx", C.c_int32 * MAXB), ("box_status", C.c_int32 * MAXB), ("omit", C.c_uint8 * MAXB), ("pad_", C.c_uint8 * 4),
                ("objects", (C.c_double * 4) * MAXB), ("box_velocity", (C.c_double * 2) * MAXB)]


class Tracker:
    """sd_tracker: System::TrackStereo / TrackRGBD / TrackMonocular for n_lanes independent camera streams, one frame per lane per call
    (Tracking::GrabImage* -> Frame::Frame -> Track_new's dynamic block -> match vs mLastFrame -> q_frame)."""

    def __init__(self, extractor, cfg, sensor, n_lanes, channels=1, rgb_order=True, track_last=True, depth_f32=False, ini_features=0, lookahead=0):
        """depth_f32: the depth images are CV_32F (Tracking.cc:271-272); ini_features: nFeatures of mpIniORBextractor for monocular lanes
        that are not initialised (Tracking.cc:127-128, 335-338), 0 = none."""
        p = TrackerParams()
        p.depth_type, p.ini_features, p.lookahead = int(bool(depth_f32)), int(ini_features), int(lookahead)
        p.sensor, p.width, p.height, p.channels, p.rgb_order = sensor, cfg["width"], cfg["height"], channels, int(bool(rgb_order))
        p.n_lanes, p.track_last = n_lanes, int(bool(track_last))
        cam = make_camera(cfg)
        for k in ("fx", "fy", "cx", "cy", "mbf", "mb", "mnMinX", "mnMaxX", "mnMinY", "mnMaxY"):
            setattr(p.cam, k, float(cam[k]))
        for k, name in enumerate(("k1", "k2", "p1", "p2", "k3")):
            p.dist[k] = float(cfg.get(name, 0.0))
        p.fps, p.depth_map_factor, p.th_depth = float(cfg["fps"]), float(cfg.get("depth_map_factor", 1.0)), float(cfg.get("th_depth", 0.0))
        self.params, self.ex, self.cfg, self.sensor, self.n_lanes = p, extractor, cfg, sensor, n_lanes
        self.images_per_lane = 2 if sensor == SENSOR_STEREO else 1
        self.h = C.c_void_p()
        check(lib().sd_tracker_create(C.byref(self.h), extractor.h, C.byref(p)))
        bh = C.c_void_p()
        check(lib().sd_tracker_batch(self.h, C.byref(bh)))
        self.batch = Batch(extractor, cfg["width"], cfg["height"], 0, handle=bh.value)
        self.results = (LaneResult * n_lanes)()

    def close(self):
        if self.h:
            self.batch.close()
            lib().sd_tracker_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(lib().sd_tracker_reset(self.h))

    def prefetch(self, d_images, stride, image_pitch, n_frames, d_depth=0, depth_stride=0, depth_pitch=0, stream=None):
        """The history-free half (cvtColor, extraction, undistortion, stereo / RGB-D association) of the next n_frames frames of every lane in one
        batch: images in frame-major order (frame k, lane s, eye e).  The following n_frames track() calls pass d_images = 0."""
        check(lib().sd_tracker_prefetch(self.h, C.c_void_p(d_images), stride, image_pitch, C.c_void_p(d_depth or 0), depth_stride, depth_pitch, n_frames,
                                        C.c_void_p(stream or 0)))

    def record_bytes(self):
        """Bytes of one prefetched-frame record (sd_tracker_prefetched_record_bytes); 0 without lookahead."""
        return int(lib().sd_tracker_prefetched_record_bytes(self.h))

    def export_prefetched(self, first_frame, n_frames, d_records, record_stride=0, stream=None):
        """Frames [first_frame, first_frame + n_frames) of the newest prefetched block -> records at d_records (record_stride bytes apart, 0 = packed),
        frame-major (frame k, lane s)."""
        check(lib().sd_tracker_export_prefetched(self.h, int(first_frame), int(n_frames), C.c_void_p(d_records), int(record_stride or self.record_bytes()),
                                                 C.c_void_p(stream or 0)))

    def import_prefetched(self, d_records, n_frames, record_stride=0, stream=None):
        """n_frames * n_lanes records (frame-major) become the next outstanding prefetched block: frames whose history-free half ran elsewhere."""
        check(lib().sd_tracker_import_prefetched(self.h, C.c_void_p(d_records), int(record_stride or self.record_bytes()), int(n_frames), C.c_void_p(stream or 0)))

    def discard_prefetched(self):
        """A worker drops its newest prefetched block after exporting it."""
        check(lib().sd_tracker_discard_prefetched(self.h))

    def set_state(self, state):
        """state: per lane, bit0 = initialised, bit1 = mState == OK && !mVelocity.empty(); None = the automatic rule."""
        if state is None:
            check(lib().sd_tracker_set_state(self.h, None))
        else:
            st = np.ascontiguousarray(state, np.int32).reshape(self.n_lanes)
            check(lib().sd_tracker_set_state(self.h, _p(st)))

    def set_mappoints(self, xw_list, flags_list):
        """The back end's MapPoints of the frames just tracked: per lane (n, 3) f32 world positions + (n,) u8 flags, or None to keep the
        lane's own stereo points."""
        S, cap = self.n_lanes, self.batch.cap
        xw = np.zeros((S, cap, 3), np.float32); fl = np.zeros((S, cap), np.uint8); n = np.full(S, -1, np.int32)
        for l in range(S):
            if xw_list[l] is None:
                continue
            k = len(flags_list[l]); n[l] = k
            xw[l, :k] = np.asarray(xw_list[l], np.float32).reshape(k, 3); fl[l, :k] = flags_list[l]
        check(lib().sd_tracker_set_mappoints(self.h, _p(xw), _p(fl), _p(n)))

    def track(self, d_images, stride, image_pitch, timestamps, boxes=None, n_boxes=None, d_depth=0, depth_stride=0, depth_pitch=0,
              Tcw=None, Twc=None, stream=None):
        """boxes: (n_lanes, MAXB, 4) f64 with n_boxes (n_lanes,) int32 (-1 = the constructor without boxes), or a list of (k, 4) arrays / None per lane."""
        S = self.n_lanes
        ts = np.ascontiguousarray(timestamps, np.float64).reshape(S)
        if boxes is not None and n_boxes is None:
            bx = np.zeros((S, MAXB, 4), np.float64); nb = np.full(S, -1, np.int32)
            for l in range(S):
                if boxes[l] is not None:
                    a = np.asarray(boxes[l], np.float64).reshape(-1, 4)
                    nb[l] = len(a); bx[l, :len(a)] = a
            boxes, n_boxes = bx, nb
        bp = _p(np.ascontiguousarray(boxes, np.float64)) if boxes is not None else None
        self._keep = (boxes, n_boxes)
        np_ = _p(np.ascontiguousarray(n_boxes, np.int32)) if n_boxes is not None else None
        tc = np.ascontiguousarray(Tcw, np.float32).reshape(S, 16) if Tcw is not None else None
        tw = np.ascontiguousarray(Twc, np.float32).reshape(S, 16) if Twc is not None else None
        check(lib().sd_tracker_track(self.h, C.c_void_p(d_images or 0), stride, image_pitch, C.c_void_p(d_depth or 0), depth_stride, depth_pitch,
                                     bp, np_, _p(ts), _p(tc) if tc is not None else None, _p(tw) if tw is not None else None,
                                     C.cast(self.results, C.c_void_p), C.c_void_p(stream or 0)))
        return self.results


class RefQueue:
    """q_frame of Tracking (Tracking.h:109) with the reference-frame choice of Track_new (Tracking.cc:620-666)."""

    def __init__(self):
        self.h = C.c_void_p()
        check(lib().sd_refqueue_create(C.byref(self.h)))

    def __del__(self):
        try:
            lib().sd_refqueue_destroy(self.h)
        except Exception:
            pass

    def __len__(self):
        n = C.c_int(); check(lib().sd_refqueue_size(self.h, C.byref(n))); return n.value

    def clear(self):
        check(lib().sd_refqueue_clear(self.h))

    def candidate(self, t, has_boxes=True):
        s = C.c_int(); check(lib().sd_refqueue_candidate(self.h, t, int(has_boxes), C.byref(s))); return s.value

    def reject(self):
        a = C.c_int(); check(lib().sd_refqueue_reject(self.h, C.byref(a))); return bool(a.value)

    def push(self, t, slot, has_boxes, max_frames):
        e = C.c_int(); check(lib().sd_refqueue_push(self.h, t, slot, int(has_boxes), max_frames, C.byref(e))); return e.value


def box_track(boxes, last_objects, last_box_idx, last_omit, last_velocity, img_cols, img_rows, cap=MAXB):
    """Frame::boxTrack (src/Frame.cc:481-552) through the C ABI (host code)."""
    n = len(boxes)
    bx = np.zeros((cap, 4), np.float64); bx[:n] = np.asarray(boxes, np.float64).reshape(n, 4)
    lo = np.ascontiguousarray(last_objects, np.float64).reshape(-1, 4)
    li = np.ascontiguousarray(last_box_idx, np.int32); lm = np.ascontiguousarray(last_omit, np.uint8)
    lv = np.ascontiguousarray(last_velocity, np.float64).reshape(-1, 2)
    idx = np.zeros(cap, np.int32); om = np.zeros(cap, np.uint8); vel = np.zeros((cap, 2), np.float64)
    n2 = C.c_int()
    check(lib().sd_box_track(_p(bx), n, cap, _p(lo), len(lo), _p(li), _p(lm), _p(lv), int(img_cols), int(img_rows), _p(idx), _p(om),
                             _p(vel), C.byref(n2)))
    m = n2.value
    return bx[:m].copy(), idx[:m].copy(), om[:m].copy(), vel[:m].copy()


def image_bounds(cols, rows, K4, dist5):
    """Frame::ComputeImageBounds (src/Frame.cc:844-872) through the C ABI -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    k = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist5, np.float32); b = np.zeros(4, np.float32)
    check(lib().sd_image_bounds(int(cols), int(rows), _p(k), _p(d), _p(b)))
    return b


def distortion_of(cfg):
    return np.array([cfg.get(k, 0.0) for k in ("k1", "k2", "p1", "p2", "k3")], np.float32)


def make_camera(cfg):
    """Frame statics: mb = mbf / fx; bounds = the image rectangle for an undistorted camera (Frame.cc:864-870), the undistorted
    corner points otherwise (ComputeImageBounds)."""
    fx = np.float32(cfg["fx"]); bf = np.float32(cfg["bf"])
    cam = dict(fx=fx, fy=np.float32(cfg["fy"]), cx=np.float32(cfg["cx"]), cy=np.float32(cfg["cy"]), mbf=bf,
               mb=np.float32(bf / fx), mnMinX=np.float32(0), mnMaxX=np.float32(cfg["width"]), mnMinY=np.float32(0),
               mnMaxY=np.float32(cfg["height"]))
    d = distortion_of(cfg)
    if d[0] != 0:
        b = image_bounds(cfg["width"], cfg["height"], [cam["fx"], cam["fy"], cam["cx"], cam["cy"]], d)
        cam.update(mnMinX=b[0], mnMaxX=b[1], mnMinY=b[2], mnMaxY=b[3])
    return cam


def camera_array(cam):
    return np.array([cam[k] for k in ("fx", "fy", "cx", "cy", "mbf", "mb", "mnMinX", "mnMaxX", "mnMinY", "mnMaxY")],
                    np.float32)


def cvt_gray_device(d_src, width, height, src_stride, src_pitch, channels, rgb_order, d_dst, dst_stride, dst_pitch,
                    n_images, stream=None):
    """cvtColor(RGB|BGR[A] -> GRAY) of Tracking::GrabImage* (src/Tracking.cc:175-200,256-269) on device buffers."""
    check(lib().sd_cvt_gray_device(C.c_void_p(d_src), width, height, src_stride, src_pitch, channels, int(rgb_order),
                                   C.c_void_p(d_dst), dst_stride, dst_pitch, n_images, C.c_void_p(stream or 0)))


def depth_to_f32_device(d_src, width, height, src_stride_elems, factor, d_dst, n_images, src_pitch_elems, stream=None):
    check(lib().sd_depth_to_f32_device(C.c_void_p(d_src), width, height, src_stride_elems, factor, C.c_void_p(d_dst),
                                       n_images, src_pitch_elems, C.c_void_p(stream or 0)))


def hamming_matrix_device(d_a, na, d_b, nb, d_out, stream=None):
    check(lib().sd_hamming_matrix_device(C.c_void_p(d_a), na, C.c_void_p(d_b), nb, C.c_void_p(d_out),
                                         C.c_void_p(stream or 0)))


FRAME_BOXES_BYTES = 16 + MAXB * 32 + 3 * MAXB * 4 + (MAXB + 1) * 4 + 4        # sizeof(sd_frame_boxes)


def batch_boxes_device(batch):
    """Device pointer of the per-slot sd_frame_boxes records."""
    p = C.c_void_p()
    check(lib().sd_batch_boxes_device(batch.h, C.byref(p)))
    return p.value


class _DevBytes:
    """Zero-copy view of library-owned HBM for torch (torch.distributed needs tensors)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def as_torch_u8(ptr, nbytes):
    import torch
    return torch.as_tensor(_DevBytes(ptr, nbytes), device="cuda")


def DescriptorDistance(a, b):
    """ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1804-1820), host popcount."""
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().sd_descriptor_distance(_p(a), _p(b))


def device_count():
    n = C.c_int()
    rc = lib().sd_device_count(C.byref(n))
    return n.value if rc == SD_OK else 0


class Vocabulary:
    """ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) as one packed buffer in HBM."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def load_text(cls, path):
        h = C.c_void_p()
        check(lib().sd_vocab_load_text(C.byref(h), str(path).encode()))
        return cls(h)

    @classmethod
    def from_nodes(cls, voc):
        """voc: dict(k, L, scoring, weighting, parent, is_leaf, desc, weight) = the node lines of the text file."""
        h = C.c_void_p()
        par = np.ascontiguousarray(voc["parent"], np.int32); leaf = np.ascontiguousarray(voc["is_leaf"], np.uint8)
        d = np.ascontiguousarray(voc["desc"], np.uint8); w = np.ascontiguousarray(voc["weight"], np.float64)
        check(lib().sd_vocab_from_nodes(C.byref(h), int(voc["k"]), int(voc["L"]), int(voc["scoring"]), int(voc["weighting"]), len(par),
                                        _p(par), _p(leaf), _p(d), _p(w)))
        return cls(h)

    @classmethod
    def from_packed_device(cls, d_ptr, nbytes):
        h = C.c_void_p()
        check(lib().sd_vocab_from_packed_device(C.byref(h), C.c_void_p(d_ptr), C.c_size_t(nbytes)))
        return cls(h)

    @staticmethod
    def packed_bytes(n_nodes):
        f = lib().sd_vocab_packed_bytes
        f.restype = C.c_size_t
        return int(f(int(n_nodes)))

    def info(self):
        v = [C.c_int() for _ in range(6)]
        check(lib().sd_vocab_info(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("k", "L", "scoring", "weighting", "n_nodes", "n_words"), [x.value for x in v]))

    def packed_device(self):
        p = C.c_void_p(); n = C.c_size_t()
        check(lib().sd_vocab_packed_device(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def nodes(self):
        n = self.info()["n_nodes"]
        par = np.zeros(n, np.int32); nch = np.zeros(n, np.int32); wid = np.zeros(n, np.int32)
        d = np.zeros((n, 32), np.uint8); w = np.zeros(n, np.float64)
        check(lib().sd_vocab_download_nodes(self.h, _p(par), _p(nch), _p(wid), _p(d), _p(w)))
        return dict(parent=par, n_children=nch, word_id=wid, desc=d, weight=w)

    def close(self):
        if self.h:
            lib().sd_vocab_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
