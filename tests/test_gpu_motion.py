"""GPU parity of the TrackHomo model fit (spec Q13): H / F from the projection matcher's point pairs, inlier masks and
the reference's choice between the two (Tracking.cc:1026-1075), against the oracle's independent restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seq(gpu, fe, synth):
    cfg = synth.KITTI_STEREO
    T = 4
    frames = [synth.stereo_frame(seq=6, t=t) for t in range(T)]
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    b = fe.Batch(ex, cfg["width"], cfg["height"], 2 * T)
    b.extract_host(np.stack([im for (l, r, _) in frames for im in (l, r)]))
    b.stereo_match(T, cfg["bf"], cfg["fx"])
    cam = fe.make_camera(cfg)
    b.assign_grid(2 * T, cam)
    I = np.eye(4, dtype=np.float32)
    b.unproject(2, T, cam, np.tile(I, (T, 1, 1)))
    yield dict(b=b, cfg=cfg, cam=cam, T=T)
    b.close()


@pytest.mark.parametrize("th,gap", [(15.0, 1), (30.0, 2)])
def test_estimate_motion(seq, fe, orc, th, gap):
    b, T, cam = seq["b"], seq["T"], seq["cam"]
    I = np.eye(4, dtype=np.float32)
    npairs = T - gap
    cur = [2 * (p + gap) for p in range(npairs)]; last = [2 * p for p in range(npairs)]
    b.search_by_projection(cur, last, np.tile(I, (npairs, 1, 1)), np.tile(I, (npairs, 1, 1)), cam, th, False, True)
    b.estimate_motion()
    for p in range(npairs):
        m, pairs, nm = b.download_matches(p)
        kl, _, _ = b.download(last[p]); kc, _, _ = b.download(cur[p])
        p1 = np.stack([kl["x"][pairs[:, 0]], kl["y"][pairs[:, 0]]], 1); p2 = np.stack([kc["x"][pairs[:, 1]], kc["y"][pairs[:, 1]]], 1)
        assert len(pairs) > 200
        o = orc.estimate_motion(p1, p2)
        g = b.download_motion(p)
        assert g["flag"] == o["flag"] and g["flag"] in (1, 2)
        assert g["n_h"] == o["n_h"] and g["n_f"] == o["n_f"] and o["n_h"] > 0.5 * len(pairs)
        assert np.array_equal(g["mask_h"], o["mask_h"]) and np.array_equal(g["mask_f"], o["mask_f"])
        for k in ("H", "F"):
            s = np.abs(o[k]).max()
            assert np.max(np.abs(g[k] - o[k])) <= 1e-9 * s, k
        assert np.max(np.abs(g["HorF"] - o["HorF"])) <= 1e-6 * np.abs(o["HorF"]).max()
        # the synthetic camera motion is a zoom of 1.01 about the principal point plus a 3 px shift per frame: H must explain it
        s = 1.01 ** gap
        Ht = np.array([[s, 0, (3.0 * gap - seq["cfg"]["cx"]) * s + seq["cfg"]["cx"]], [0, s, -seq["cfg"]["cy"] * s + seq["cfg"]["cy"]], [0, 0, 1]])
        q = (g["H"] @ np.c_[p1, np.ones(len(p1))].T).T; q = q[:, :2] / q[:, 2:]
        qt = (Ht @ np.c_[p1, np.ones(len(p1))].T).T; qt = qt[:, :2] / qt[:, 2:]
        assert np.median(np.linalg.norm(q - qt, axis=1)) < 1.0


def test_estimate_motion_below_the_checkpoints(gpu, fe, orc, synth):
    """Pairs matched between frames of two DIFFERENT scenes through a wide window: every match is false, the best inlier ratios stay under
    the checkpoint thresholds (0.53 N for H after 64 hypotheses, 0.66 N for F after 128), and the fit runs all 512 + 1024 hypotheses
    (stage 1 of k_motion_models / k_motion_count).  GPU == oracle as in the well-conditioned case."""
    cfg = synth.KITTI_STEREO
    frames = [synth.stereo_frame(seq=6, t=0), synth.stereo_frame(seq=9, t=0), synth.stereo_frame(seq=12, t=1)]
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    b = fe.Batch(ex, cfg["width"], cfg["height"], 6)
    try:
        b.extract_host(np.stack([im for (l, r, _) in frames for im in (l, r)]))
        b.stereo_match(3, cfg["bf"], cfg["fx"])
        cam = fe.make_camera(cfg)
        b.assign_grid(6, cam)
        I = np.eye(4, dtype=np.float32)
        b.unproject(2, 3, cam, np.tile(I, (3, 1, 1)))
        cur, last = [2, 4], [0, 2]
        b.search_by_projection(cur, last, np.tile(I, (2, 1, 1)), np.tile(I, (2, 1, 1)), cam, 40.0, False, True)
        b.estimate_motion()
        below_h = below_f = 0
        for p in range(2):
            m, pairs, nm = b.download_matches(p)
            assert len(pairs) >= 20, "the wide window must leave enough (false) matches for a fit"
            kl, _, _ = b.download(last[p]); kc, _, _ = b.download(cur[p])
            p1 = np.stack([kl["x"][pairs[:, 0]], kl["y"][pairs[:, 0]]], 1); p2 = np.stack([kc["x"][pairs[:, 1]], kc["y"][pairs[:, 1]]], 1)
            o = orc.estimate_motion(p1, p2)
            g = b.download_motion(p)
            assert g["flag"] == o["flag"] and g["n_h"] == o["n_h"] and g["n_f"] == o["n_f"]
            assert np.array_equal(g["mask_h"], o["mask_h"]) and np.array_equal(g["mask_f"], o["mask_f"])
            for k in ("H", "F"):
                sc = max(np.abs(o[k]).max(), 1e-30)
                assert np.max(np.abs(g[k] - o[k])) <= 1e-9 * sc, k
            below_h += o["n_h"] < 0.53 * len(pairs)
            below_f += o["n_f"] < 0.66 * len(pairs)
        assert below_h > 0 and below_f > 0, "the case must exercise the full hypothesis sets"
    finally:
        b.close()
