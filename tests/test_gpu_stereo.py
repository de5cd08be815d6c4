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


def test_stereo_dense_band_takes_the_multi_pass_branches(gpu, fe, orc, synth):
    """All texture of a 752 x 160 pair sits in a 36-row band and the extractor is asked for 4000 features: a 16-row chunk of k_stereo_match then
    holds far more than SD_SR_LEFT (128) left key points and its row band more than SD_SR_CAND (256) right candidates, so the `lb` / `cb` pass
    loops (csrc/k_frame.h) run several times and `s_best` carries (distance, iR) across candidate passes -- the branches the KITTI / TUM
    geometries never reach.  Bit-exact against the oracle's Frame::ComputeStereoMatches (/root/reference/src/Frame.cc:874-1048)."""
    cfg = synth.KITTI_STEREO
    W, H = 752, 160
    rng = np.random.Generator(np.random.PCG64(20260104))
    left = np.full((H, W + 16), 128, np.uint8)
    for y in range(62, 98, 3):                          # 3 x 3 blocks of random intensity: corners everywhere in the band
        for x in range(0, W + 16, 3):
            left[y:y + 3, x:x + 3] = rng.integers(20, 236)
    right = np.ascontiguousarray(left[:, 9:9 + W])      # disparity 9 px
    left = np.ascontiguousarray(left[:, :W])
    NL = 5                                              # 160 rows: the sixth level would have no FAST cell left (spec Q8: unsupported geometry)
    ex = fe.ORBextractor(4000, cfg["scale_factor"], NL, cfg["ini_th_fast"], cfg["min_th_fast"])
    b = fe.Batch(ex, W, H, 2)
    try:
        b.extract_host(np.stack([left, right]))
        b.stereo_match(1, cfg["bf"], cfg["fx"])
        oL = orc.Extractor(4000, cfg["scale_factor"], NL, cfg["ini_th_fast"], cfg["min_th_fast"])
        oR = orc.Extractor(4000, cfg["scale_factor"], NL, cfg["ini_th_fast"], cfg["min_th_fast"])
        kL, dL = oL(left); kR, dR = oR(right)
        kpg, descg, _ = b.download(0)
        assert kpg.tobytes() == kL.tobytes() and np.array_equal(descg, dL), "left extraction"
        rows = np.floor(kL["y"]).astype(int)
        chunk = np.bincount(rows // 16, minlength=H // 16 + 1)
        assert chunk.max() > 2 * 128, "a 16-row chunk must hold several passes of left key points (has %d)" % chunk.max()
        rrows = np.floor(kR["y"]).astype(int)
        c = int(np.argmax(chunk))
        band = int(((rrows >= 16 * c - 9) & (rrows <= 16 * c + 24)).sum())
        assert band > 2 * 256, "the chunk's row band must hold several passes of right candidates (has %d)" % band
        our, odep, osad, nm = orc.stereo_matches(oL, oR, kL, dL, kR, dR, cfg["bf"], cfg["fx"])
        ur, dep, sad = b.download_stereo(0)
        n = len(kL)
        assert nm > 300, "the shifted pair must match (%d)" % nm
        assert np.array_equal(sad[:n], osad), "SAD distances"
        assert np.array_equal(ur[:n].view(np.uint32), our.view(np.uint32)), "mvuRight"
        assert np.array_equal(dep[:n].view(np.uint32), odep.view(np.uint32)), "mvDepth"
    finally:
        b.close()
