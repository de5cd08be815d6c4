"""GPU parity of the dynamic-object cull: firstSeparate, Separate (BF cross-check + classifyH/F), UpdateFrame."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(synth, orc, cfg, seq, ts):
    out = []
    for t in ts:
        rgb, depth, _ = synth.rgbd_frame(seq, t, cfg)
        out.append((orc.cvt_gray(rgb, 1), depth))
    return out


def _boxes(synth, cfg, seq, t, extra):
    rows = synth.boxes_for_frame(seq, t, cfg)
    rects = synth.rows_to_rects(rows)
    return np.concatenate([rects, extra]) if len(extra) else rects


@pytest.fixture(scope="module", params=[0, 5000], ids=["settings", "5000-features-one-box-over-2048"])
def scene(request, gpu, fe, orc, synth):
    """params: 0 = the settings file's nFeatures (2000); 5000 = an extractor that yields more than 2048 key points per image AND a box that holds more
    than 2048 of them in both frames -- the reference has no bound here (Frame.cc:555-604 and Tracking.cc:1093-1239 work on std::vectors); rounds 1-3
    refused it (SD_ERR_UNSUPPORTED), since round 4 the key-point tables of k_box_separate / k_separate are sized by the workspace and a box's train
    descriptors pass through LDS in chunks."""
    import torch
    cfg = dict(synth.KITTI03_RGBD)
    big = request.param > 0
    if big:
        cfg["n_features"] = request.param
    ts = [0, 3]
    fr = _frames(synth, orc, cfg, 8, ts)
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    b = fe.Batch(ex, cfg["width"], cfg["height"], 3)
    b.extract_host(np.stack([g for g, _ in fr]))
    factor = float(np.float32(1.0) / np.float32(cfg["depth_map_factor"]))
    d_dev = torch.from_numpy(np.stack([d for _, d in fr]).view(np.int16)).cuda()
    b.rgbd_from_u16(d_dev.data_ptr(), cfg["width"], cfg["width"] * cfg["height"], 2, factor, cfg["bf"])
    # boxes: the 3 synthetic ones + one EMPTY box (flat border region has no corners inside a 2x2 box) in the
    # middle of the list (triggers the erase quirk) + one box overlapping box 0 (keypoints in two boxes)
    per_frame = []
    for t in ts:
        base = _boxes(synth, cfg, 8, t, np.zeros((0, 4)))
        empty = np.array([[0.25, 0.25, 1.5, 1.5]])
        overlap = base[0:1] + np.array([[20., 10., 0., 0.]])
        boxes = np.concatenate([base[:1], empty, base[1:], overlap] + ([np.array([[60., 30., 1100., 320.]])] if big else []))
        idx = np.arange(len(boxes), dtype=np.int32) + 10
        per_frame.append((boxes, idx))
    # oracle side
    ref = []
    for (g, d), (boxes, idx) in zip(fr, per_frame):
        o = orc.Extractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
        kp, desc = o(g)
        ur, dep = orc.stereo_from_rgbd(kp, orc.depth_to_f32(d, factor), cfg["bf"])
        r = orc.first_separate(kp, desc, boxes, idx, np.zeros(len(boxes), np.uint8), np.zeros((len(boxes), 2)))
        r["ur"] = ur[r["perm"]]; r["dep"] = dep[r["perm"]]
        ref.append(r)
    b.first_separate([0, 1], [p[0] for p in per_frame], [p[1] for p in per_frame])
    if big:
        for r in ref:
            assert r["Ns"] + r["Nd"] > 2048 and np.diff(r["boxStart"]).max() > 2048, "the big case must put more than 2048 key points into one box (%d, %d)" % (r["Ns"] + r["Nd"], np.diff(r["boxStart"]).max())
    yield dict(b=b, cfg=cfg, ref=ref, ts=ts)
    b.close()


def test_first_separate(scene, fe):
    b = scene["b"]
    for slot, r in enumerate(scene["ref"]):
        g = b.download_boxes(slot)
        assert r["Nd"] > 50, "the synthetic boxes must contain keypoints"
        assert g["n_static"] == r["Ns"] and g["n_all"] == r["Ns"] + r["Nd"]
        assert g["nb"] == len(r["boxes"]) and np.array_equal(g["boxes"], r["boxes"]) and np.array_equal(g["box_idx"], r["box_idx"])
        # device lists index the dynamic arrays; the oracle's index the concatenated (static ++ dynamic) arrays
        assert np.array_equal(g["boxStart"], r["boxStart"]) and np.array_equal(g["boxItems"] + r["Ns"], r["boxItems"])
        assert (g["box_status"] == -1).all()
        kp, desc, _ = b.download(slot)
        Ns = r["Ns"]
        assert len(kp) == Ns and kp.tobytes() == r["kp"][:Ns].tobytes() and np.array_equal(desc, r["desc"][:Ns])   # N = N_s
        ur, dep = b.download_rgbd(slot)
        assert np.array_equal(ur[:Ns].view(np.uint32), r["ur"][:Ns].view(np.uint32))
        assert np.array_equal(dep[:Ns].view(np.uint32), r["dep"][:Ns].view(np.uint32))
        dk, dd, dur, ddep = b.download_dynamic(slot)
        assert dk.tobytes() == r["kp"][Ns:].tobytes(), "dynamic keypoints (class_id = original index)"
        assert np.array_equal(dd, r["desc"][Ns:])
        assert np.array_equal(dur.view(np.uint32), r["ur"][Ns:].view(np.uint32)) and np.array_equal(ddep.view(np.uint32), r["dep"][Ns:].view(np.uint32))
        # a keypoint inside two boxes is listed in both
        items = g["boxItems"]
        assert len(items) > len(np.unique(items))


def _similarity_H(cfg, dt):
    # synth.cut_frame: x_t = (x_0 + 3t - cx) * s^t + cx  ->  x_cur = a * x_ref + bx (ref = t0, cur = t0 + dt)
    s = 1.01 ** dt
    a = s
    bx = (3.0 * dt - cfg["cx"]) * s + cfg["cx"]      # for t0 = 0
    by = (-cfg["cy"]) * s + cfg["cy"]
    return np.array([[a, 0, bx], [0, a, by], [0, 0, 1]], np.float32)


@pytest.mark.parametrize("flag", [1, 2])
def test_separate_and_update_frame(scene, fe, orc, flag):
    b, cfg, ref = scene["b"], scene["cfg"], scene["ref"]
    cur, rf = ref[1], ref[0]
    if flag == 1:
        M = _similarity_H(cfg, scene["ts"][1] - scene["ts"][0])           # points_cur = H21 * points_ref
    else:
        # a fundamental matrix compatible with that similarity: F = [e]x H with the epipole at the principal point
        H = _similarity_H(cfg, scene["ts"][1] - scene["ts"][0]).astype(np.float64)
        e = np.array([cfg["cx"], cfg["cy"], 1.0])
        ex = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
        M = (ex @ H).astype(np.float32)
    last_idx = [np.array([10, 12, 13], np.int32)]
    last_status = [np.array([0, 2, -1], np.int32)]
    cur_boxes = b.download_boxes(1)
    oret, osc, ods, odyn, omt = orc.separate(M, flag, dict(kp=cur["kp"], desc=cur["desc"], boxStart=cur["boxStart"],
                                                           boxItems=cur["boxItems"], box_idx=cur["box_idx"]),
                                             dict(kp=rf["kp"], desc=rf["desc"], boxStart=rf["boxStart"], boxItems=rf["boxItems"],
                                                  box_idx=rf["box_idx"]),
                                             last_idx[0], last_status[0], cur_boxes["box_status"])
    b.separate([1], [0], M[None], [flag], last_idx, last_status)
    ret, ds, dyn, mt = b.download_separate(0)
    nb = len(cur["box_idx"])
    assert len(omt) > 20, "boxes should produce cross-checked matches"
    assert ret == oret
    assert np.array_equal(ds[:nb + 1], ods) and np.array_equal(mt, omt) and np.array_equal(dyn, odyn)
    assert np.array_equal(b.download_boxes(1)["box_status"], osc)
    # UpdateFrame
    n_before = int(b.counts(2)[1])
    app = orc.update_frame(cur["kp"], cur["boxStart"], cur["boxItems"], ods, odyn)
    b.update_frame(only_if_static=False)
    n_after = int(b.counts(2)[1])
    assert n_after == n_before + len(app)
    kp, desc, _ = b.download(1)
    exp_kp = np.concatenate([cur["kp"][:cur["Ns"]], cur["kp"][app]]); exp_desc = np.concatenate([cur["desc"][:cur["Ns"]], cur["desc"][app]])
    assert kp.tobytes() == exp_kp.tobytes() and np.array_equal(desc, exp_desc)
    ur, dep = b.download_rgbd(1)
    assert np.array_equal(ur[:n_after].view(np.uint32), np.concatenate([cur["ur"][:cur["Ns"]], cur["ur"][app]]).view(np.uint32))
    # restore N = N_s for the next parametrisation (the dynamic arrays and box lists are untouched by UpdateFrame)
    _truncate(b, fe, 1, cur["Ns"])


def _truncate(b, fe, slot, n):
    import torch
    kp_p, desc_p, cnt_p, cap = b.results_device()
    t = fe.as_torch_u8(cnt_p + 4 * slot, 4)
    t.copy_(torch.from_numpy(np.array([n], np.int32).view(np.uint8)).cuda())
    torch.cuda.synchronize()


def test_randomised_cull_configurations(gpu):
    """6 fixed draws of tools/fuzz_cull.py (random box sets incl. empty / overlapping / partly outside boxes, H or F exact or
    perturbed, random carried-over states): firstSeparate, Separate and UpdateFrame identical to the oracle.  (70 draws were
    run when written: 26 of 34 Separate calls re-admitted their boxes, 25 box-status changes, 19 empty boxes dropped.)"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_cull", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_cull.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert m.run(6, 17) == 0
