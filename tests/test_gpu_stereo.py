"""GPU parity of Frame::ComputeStereoMatches / ComputeStereoFromRGBD against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_stereo_matches_oracle(gpu, fe, orc, synth):
    cfg = synth.KITTI_STEREO
    frames = [synth.stereo_frame(seq=2, t=t) for t in range(3)]
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    b = fe.Batch(ex, cfg["width"], cfg["height"], 2 * len(frames))
    b.extract_host(np.stack([im for (l, r, _) in frames for im in (l, r)]))
    b.stereo_match(len(frames), cfg["bf"], cfg["fx"])
    for f, (l, r, _) in enumerate(frames):
        oL = orc.Extractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
        oR = orc.Extractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
        kL, dL = oL(l); kR, dR = oR(r)
        our, odep, osad, nm = orc.stereo_matches(oL, oR, kL, dL, kR, dR, cfg["bf"], cfg["fx"])
        ur, dep, sad = b.download_stereo(f)
        n = len(kL)
        assert nm > 100, "synthetic pair should produce stereo matches"
        assert np.array_equal(sad[:n], osad), "SAD distances, frame %d" % f
        assert np.array_equal(ur[:n].view(np.uint32), our.view(np.uint32)), "mvuRight, frame %d" % f
        assert np.array_equal(dep[:n].view(np.uint32), odep.view(np.uint32)), "mvDepth, frame %d" % f
    b.close()


def test_rgbd_depth_lookup_matches_oracle(gpu, fe, orc, synth):
    import torch
    cfg = synth.KITTI03_RGBD
    rgb, depth, _ = synth.rgbd_frame(seq=3, t=0)
    gray = orc.cvt_gray(rgb, 1)
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    b = fe.Batch(ex, cfg["width"], cfg["height"], 1)
    b.extract_host(gray[None])
    factor = np.float32(1.0) / np.float32(cfg["depth_map_factor"])
    d_dev = torch.from_numpy(depth.astype(np.int16)).cuda()
    b.rgbd_from_u16(d_dev.data_ptr(), cfg["width"], cfg["width"] * cfg["height"], 1, float(factor), cfg["bf"])
    ur, dep = b.download_rgbd(0)
    o = orc.Extractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    kp, _ = o(gray)
    our, odep = orc.stereo_from_rgbd(kp, orc.depth_to_f32(depth, float(factor)), cfg["bf"])
    n = len(kp)
    assert (odep > 0).sum() > 100
    assert np.array_equal(ur[:n].view(np.uint32), our.view(np.uint32))
    assert np.array_equal(dep[:n].view(np.uint32), odep.view(np.uint32))
    b.close()
