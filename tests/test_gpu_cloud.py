"""GPU parity of the dense RGB-D back-projection with the dynamic mask: PointCloudMapping::generatePointCloud
(src/pointcloudmapping.cc:59-103) fed by the dyn_boxes filter of Tracking::CreateNewKeyFrame (src/Tracking.cc:1999-2007), on frames
that went through the whole TUM3 chain (BASELINE configs[3]) so that box_status holds real 0 / 2 / -1 values."""
import importlib.util
import os

import numpy as np
import pytest

import __graft_entry__ as graft

pytestmark = pytest.mark.gpu


def _load(name, fn):
    spec = importlib.util.spec_from_file_location(name, os.path.join(graft.ROOT, "oracle", fn))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_backproject_dense_tum3(gpu, fe, orc, synth):
    import torch
    P, CO = _load("sd_oracle_pipeline_c", "pipeline.py"), _load("sd_cloud_oracle", "cloud_oracle.py")
    cfg = synth.TUM3
    W, H, S, T = cfg["width"], cfg["height"], 2, 9
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    trk = fe.Tracker(ex, cfg, fe.SENSOR_RGBD, S, channels=3)
    oracles = [P.SequenceOracle(orc, cfg, P.SENSOR_RGBD) for _ in range(S)]
    cam = fe.make_camera(cfg)
    factor = float(np.float32(1.0) / np.float32(cfg["depth_map_factor"]))
    cap_pts = ((W + 2) // 3) * ((H + 2) // 3)
    d_pts = torch.zeros((S, cap_pts, 16), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((S, 2), dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(5)
    Twc = np.stack([np.eye(4), np.eye(4)])
    a = 0.3
    Twc[1, :3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    Twc[1, :3, 3] = [0.5, -0.25, 2.0]
    seen_status = set()
    try:
        for t in range(T):
            fr = [synth.rgbd_frame_dyn(51 + l, t, cfg) for l in range(S)]
            rows = [synth.boxes_for_frame(51 + l, t, cfg) for l in range(S)]
            boxes = [synth.rows_to_rects(r) for r in rows]
            masks = [synth.mask_from_boxes(r, W, H) for r in rows]
            d_rgb = torch.from_numpy(np.stack([f[0] for f in fr])).cuda()
            d_dep = torch.from_numpy(np.stack([f[1] for f in fr]).view(np.int16)).cuda()
            d_msk = torch.from_numpy(np.stack(masks)).cuda()
            res = trk.track(d_rgb.data_ptr(), W * 3, W * H * 3, [t / 30.0] * S, boxes=boxes, d_depth=d_dep.data_ptr(), depth_stride=W, depth_pitch=W * H)
            Fs = [oracles[l].track(fr[l][0], fr[l][1], boxes[l], t / 30.0) for l in range(S)]
            if t < T - 2:
                continue
            for use_mask in (True, False):
                trk.batch.backproject_dense([res[l].cur_slot for l in range(S)], d_rgb.data_ptr(), W * 3, W * H * 3, d_dep.data_ptr(), W, W * H, factor,
                                            d_msk.data_ptr() if use_mask else 0, W, W * H, cam, Twc, d_pts.data_ptr(), cap_pts, d_cnt.data_ptr())
                torch.cuda.synchronize()
                cnt = d_cnt.cpu().numpy(); pts = d_pts.cpu().numpy()
                for l in range(S):
                    F = Fs[l]
                    seen_status.update(int(s) for s in F.box_status)
                    ref, masked = CO.generate_point_cloud(fr[l][0], fr[l][1], factor, masks[l] if use_mask else None,
                                                          CO.dyn_boxes(F.objects, F.box_status), cam["fx"], cam["fy"], cam["cx"], cam["cy"], Twc[l])
                    assert (int(cnt[l, 0]), int(cnt[l, 1])) == (len(ref), masked), "frame %d lane %d: cloud size / masked_num %r vs %r" % (t, l, cnt[l], (len(ref), masked))
                    got = pts[l, :len(ref)].reshape(-1).view(fe.CLOUD_POINT_DTYPE)
                    assert got.tobytes() == ref.tobytes(), "frame %d lane %d: points differ" % (t, l)
                    if use_mask:
                        assert masked > 0 and len(ref) > 1000, "the mask must remove pixels and the depth window must keep some"
    finally:
        trk.close()
    assert (2 in seen_status or 0 in seen_status), "the chain must have produced dynamic boxes: %r" % seen_status
