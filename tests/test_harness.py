"""The example drivers' file grammars and statistics: rgbd_my.cc:133-146, 196-253; stereo_kitti.cc:173-208."""
import os
import numpy as np
import pytest


def test_box_file_grammar(pkg, synth):
    h = pkg.harness
    txt = "0 100.5 50.25 80 40\n\n1 10 5 60 50\n2 1200.0 370.0 100 100\n"
    r = h.parse_box_text(txt)
    # Rect2d(MAX(cx - w/2, 0), MAX(cy - h/2, 0), w, h): the clamp moves the corner but keeps the size (rgbd_my.cc:246-249)
    assert r.tolist() == [[60.5, 30.25, 80, 40], [0.0, 0.0, 60, 50], [1150.0, 320.0, 100, 100]]
    rows = synth.boxes_for_frame(3, 5, synth.KITTI03_RGBD)
    assert np.array_equal(h.parse_box_text(h.format_box_rows(rows)), synth.rows_to_rects(rows))
    assert h.parse_box_text("").shape == (0, 4)


def test_timing_summary_is_the_drivers(pkg):
    s = pkg.harness.timing_summary([0.5, 0.1, 0.3, 0.2])
    assert s["median"] == 0.3 and abs(s["mean"] - 0.275) < 1e-15 and s["n"] == 4        # sorted[n/2], total/n


def test_sequence_layout_round_trip(pkg, synth, tmp_path):
    h = pkg.harness
    cfg = synth.KITTI03_RGBD
    h.write_synthetic_kitti_rgbd(str(tmp_path), synth, 7, 3, cfg)
    assert h.load_times(tmp_path / "times.txt") == [0.0, 0.1, 0.2]
    frames = h.kitti_rgbd_layout(str(tmp_path), 3)
    assert os.path.basename(frames[2]["mask"]) == "mask_000002.png" and os.path.basename(frames[1]["depth"]) == "000001.png"
    for t, fr in enumerate(frames):
        rgb, depth, _ = synth.rgbd_frame(7, t, cfg)
        im = h.imread_unchanged(fr["rgb"])
        assert np.array_equal(im[:, :, ::-1], rgb)                                      # cv::imread gives BGR
        assert h.imread_unchanged(fr["depth"]).dtype == np.uint16 and np.array_equal(h.imread_unchanged(fr["depth"]), depth)
        m = h.mask_to_f32(h.imread_unchanged(fr["mask"]))
        assert m.dtype == np.float32 and set(np.unique(m)) <= {0.0, 255.0} and (m > 0).any()
        assert np.array_equal(fr["boxes"], synth.rows_to_rects(synth.boxes_for_frame(7, t, cfg)))
    st = h.kitti_stereo_layout("/data/kitti/03", 2)
    assert st[1] == dict(left="/data/kitti/03/image_2/000001.png", right="/data/kitti/03/image_3/000001.png")


@pytest.mark.gpu
def test_run_rgbd_sequence_end_to_end(gpu, pkg, fe, synth, tmp_path):
    """A synthetic sequence in the reference's layout through the per-frame caller.  The synthetic boxes are rectangles over
    the rigid background (nothing is painted into them), so once a reference frame more than 0.2 s old exists every box is
    found static: TrackHomo succeeds, Separate returns 1, UpdateFrame re-admits the box keypoints and the statuses stay -1
    (the static branch's `box_status == 1;` is a no-op in the reference, Tracking.cc:1199).  Dynamic classification itself
    is covered with crafted motion in test_gpu_cull.py."""
    h = pkg.harness
    cfg = synth.KITTI03_RGBD
    n = 8
    h.write_synthetic_kitti_rgbd(str(tmp_path), synth, 7, n, cfg)
    front = h.DynamicFrontEnd(fe, cfg, rgb_order=False)        # imread delivers BGR
    res, timing = h.run_rgbd_sequence(front, str(tmp_path), n)
    front.close()
    assert timing["n"] == n and 0 < timing["median"] < 5.0
    assert all(r["n_boxes"] == 3 for r in res)
    assert res[0]["flag"] == 0 and res[1]["flag"] == 0 and res[2]["flag"] == 0          # no frame more than 0.2 s older yet
    later = [r for r in res[3:]]
    assert all(r["flag"] in (1, 2) for r in later), [r["flag"] for r in res]
    assert all(r["matches"] >= 200 for r in later)
    assert all(r["separate_ret"] == 1 for r in later)
    assert all(r["n_keypoints"] > r["n_static"] + 20 for r in later), [(r["n_static"], r["n_keypoints"]) for r in later]
    assert all(r["n_keypoints"] == r["n_static"] for r in res[:3])
    # ids are carried from frame to frame by boxTrack
    assert all(sorted(r["box_idx"].tolist()) == [0, 1, 2] for r in res)
    assert all((r["box_status"] == -1).all() for r in res)
