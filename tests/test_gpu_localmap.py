"""GPU parity of Tracking::SearchLocalPoints: Frame::isInFrustum (Frame.cc:677-733) +
ORBmatcher::SearchByProjection(Frame, vector<MapPoint*>, th) (ORBmatcher.cc:45-129), bit-exact against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seq(gpu, fe, orc, synth):
    cfg = synth.KITTI_STEREO
    T = 4
    frames = [synth.stereo_frame(seq=9, t=t) for t in range(T)]
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    b = fe.Batch(ex, cfg["width"], cfg["height"], 2 * T)
    b.extract_host(np.stack([im for (l, r, _) in frames for im in (l, r)]))
    b.stereo_match(T, cfg["bf"], cfg["fx"])
    cam = fe.make_camera(cfg)
    b.assign_grid(2 * T, cam)
    cam10 = fe.camera_array(cam)
    ref = []
    I = np.eye(4, dtype=np.float32)
    for t in range(T):
        kp, desc, _ = b.download(2 * t)
        ur, dep, _ = b.download_stereo(t)
        xw, valid = orc.unproject(kp, dep[:len(kp)], cam10, I)
        ref.append(dict(kp=kp, desc=desc, ur=ur[:len(kp)], depth=dep[:len(kp)], xw=xw, valid=valid))
    sf = np.ones(cfg["n_levels"], np.float32)
    for l in range(1, cfg["n_levels"]): sf[l] = sf[l - 1] * np.float32(cfg["scale_factor"])     # mvScaleFactor (ORBextractor.cc:419-425)
    yield dict(b=b, cfg=cfg, cam=cam, cam10=cam10, ref=ref, T=T, sf=sf)
    b.close()


def make_local_map(orc, ref, frames, rng, obs_frac, bad_frac, flip_bits):
    """Local map = the stereo points of `frames` (world = camera frame of those frames, identity poses)."""
    pts, descs = [], []
    for t in frames:
        r = ref[t]
        idx = np.nonzero(r["valid"])[0]
        m = np.zeros(len(idx), orc.MAP_POINT_DTYPE)
        xw = r["xw"][idx]
        m["xw"] = xw
        d = np.linalg.norm(xw.astype(np.float64), axis=1)
        nrm = xw / np.maximum(d, 1e-6)[:, None] + rng.normal(scale=0.25, size=xw.shape)      # roughly facing the camera
        nrm /= np.linalg.norm(nrm, axis=1)[:, None]
        m["normal"] = nrm.astype(np.float32)
        lvl = r["kp"]["octave"][idx].astype(np.float64)
        # MapPoint::UpdateNormalAndDepth: mfMaxDistance = dist * scale[level]; mfMinDistance = mfMaxDistance / scale[nLevels-1]
        m["max_distance"] = (d * 1.2 ** lvl * rng.uniform(0.8, 1.25, len(idx))).astype(np.float32)
        m["min_distance"] = (m["max_distance"] / np.float32(1.2 ** 7)).astype(np.float32)
        fl = (rng.random(len(idx)) >= bad_frac).astype(np.uint32)
        fl |= (rng.random(len(idx)) < obs_frac).astype(np.uint32) << 1
        m["flags"] = fl
        dd = r["desc"][idx].copy()
        nflip = rng.integers(0, flip_bits + 1, len(idx))
        for k in range(len(idx)):
            for bit in rng.integers(0, 256, nflip[k]):
                dd[k, bit >> 3] ^= np.uint8(1 << (bit & 7))
        pts.append(m); descs.append(dd)
    pts = np.concatenate(pts); descs = np.concatenate(descs)
    perm = rng.permutation(len(pts))
    return pts[perm], descs[perm]


@pytest.mark.parametrize("th,obs_frac,occ_frac,flip", [(1.0, 1.0, 0.0, 20), (3.0, 1.0, 0.3, 40), (5.0, 0.7, 0.1, 60), (12.0, 0.9, 0.0, 60)])
def test_search_local_map(seq, fe, orc, th, obs_frac, occ_frac, flip):
    """Two frames per call, each against its own local map (points of the other frames).  th = 12 makes windows of
    more than 64 keypoints (the kept-64 path) and many contested keypoints (the serial replay)."""
    import torch
    b, T, cam, cam10, ref, sf = seq["b"], seq["T"], seq["cam"], seq["cam10"], seq["ref"], seq["sf"]
    rng = np.random.default_rng(int(th * 10 + obs_frac * 100))
    targets = [3, 1]
    maps = [make_local_map(orc, ref, [0, 1, 2], rng, obs_frac, 0.1, flip), make_local_map(orc, ref, [0, 2, 3], rng, obs_frac, 0.1, flip)]
    off = np.cumsum([0] + [len(m[0]) for m in maps]).astype(np.int32)
    pts = np.concatenate([m[0] for m in maps]); descs = np.concatenate([m[1] for m in maps])
    Tcw = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    for f in range(2):
        a = 0.01 * (f + 1)
        Tcw[f, :3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        Tcw[f, :3, 3] = np.array([0.05, -0.02, -0.6 * (f + 1)], np.float32)
    occ = (rng.random((2, b.cap)) < occ_frac).astype(np.uint8)
    d_pts = torch.from_numpy(pts.view(np.uint8).reshape(-1)).cuda()
    d_desc = torch.from_numpy(descs).cuda()
    d_occ = torch.from_numpy(occ).cuda()
    d_track = torch.zeros(len(pts) * 24, dtype=torch.uint8, device="cuda")
    d_pm = torch.zeros(len(pts), dtype=torch.int32, device="cuda")
    d_km = torch.zeros((2, b.cap), dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(2, dtype=torch.int32, device="cuda")
    b.search_local_map([2 * t for t in targets], off, d_pts.data_ptr(), d_desc.data_ptr(), Tcw, cam, th, 0.8, d_track.data_ptr(),
                       d_pm.data_ptr(), d_km.data_ptr(), d_nm.data_ptr(), d_occupied=d_occ.data_ptr())
    b.sync()
    track = d_track.cpu().numpy().view(orc.TRACK_DTYPE); pm = d_pm.cpu().numpy(); km = d_km.cpu().numpy(); nm = d_nm.cpu().numpy()
    total = 0
    for f, t in enumerate(targets):
        r = ref[t]
        n = len(r["kp"])
        otr, opm, okm, onm = orc.search_local_map(r["kp"], r["desc"], r["ur"], pts[off[f]:off[f + 1]], descs[off[f]:off[f + 1]], Tcw[f], cam10,
                                                  sf, th, 0.8, 0.5, occupied=occ[f, :n])
        g = track[off[f]:off[f + 1]]
        assert np.array_equal(g["in_view"], otr["in_view"]), "mbTrackInView, frame %d" % f
        assert otr["in_view"].sum() > 500
        for name in ("proj_x", "proj_y", "proj_xr", "view_cos"):
            assert np.array_equal(g[name].view(np.uint32), otr[name].view(np.uint32)), name
        assert np.array_equal(g["level"], otr["level"])
        assert nm[f] == onm, "nmatches frame %d: %d vs %d" % (f, nm[f], onm)
        assert np.array_equal(pm[off[f]:off[f + 1]], opm), "per-point matches, frame %d" % f
        assert np.array_equal(km[f, :n], okm), "F.mvpMapPoints, frame %d" % f
        total += onm
    assert total > 300, "expected plenty of matches, got %d" % total
