"""GPU parity of the GrabImage* preprocessing kernels and the Hamming primitive."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("channels,rgb", [(3, 1), (3, 0), (4, 1), (4, 0)])
def test_cvt_gray(gpu, fe, orc, channels, rgb):
    import torch
    rng = np.random.default_rng(channels * 2 + rgb)
    n, h, w = 3, 123, 517          # odd sizes: exercises the 4-pixel tail
    src = rng.integers(0, 256, (n, h, w, channels), dtype=np.uint8)
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros((n, h, w), dtype=torch.uint8, device="cuda")
    fe.cvt_gray_device(d_src.data_ptr(), w, h, w * channels, w * h * channels, channels, rgb, d_dst.data_ptr(), w, w * h, n)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], orc.cvt_gray(src[i], rgb))


def test_depth_to_f32(gpu, fe, orc):
    import torch
    rng = np.random.default_rng(9)
    d = rng.integers(0, 65536, (2, 97, 203), dtype=np.uint16)
    factor = float(np.float32(1.0) / np.float32(5000.0))
    d_src = torch.from_numpy(d.view(np.int16)).cuda()
    d_dst = torch.zeros((2, 97, 203), dtype=torch.float32, device="cuda")
    fe.depth_to_f32_device(d_src.data_ptr(), 203, 97, 203, factor, d_dst.data_ptr(), 2, 97 * 203)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i].view(np.uint32), orc.depth_to_f32(d[i], factor).view(np.uint32))


def test_hamming_matrix(gpu, fe, orc):
    import torch
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (37, 32), dtype=np.uint8); b = rng.integers(0, 256, (131, 32), dtype=np.uint8)
    a[3] = b[5]                      # distance 0
    a[4] = ~b[6]                     # distance 256
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = torch.zeros((37, 131), dtype=torch.int16, device="cuda")
    fe.hamming_matrix_device(da.data_ptr(), 37, db.data_ptr(), 131, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.int32)
    ref = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2)
    assert np.array_equal(got, ref)
    assert got[3, 5] == 0 and got[4, 6] == 256
    assert fe.DescriptorDistance(a[0], b[0]) == ref[0, 0] == orc.descriptor_distance(a[0], b[0])


def test_undistort_points_and_bounds(gpu, fe, orc):
    """Frame::UndistortKeyPoints / ComputeImageBounds with TUM1.yaml's distortion (Examples/RGB-D/TUM1.yaml): bit-exact vs the oracle."""
    import ctypes as C
    import torch
    K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)
    D = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    rng = np.random.default_rng(2)
    pts = np.stack([rng.uniform(0, 640, 5000), rng.uniform(0, 480, 5000)], 1).astype(np.float32)
    d_in = torch.from_numpy(pts).cuda(); d_out = torch.zeros_like(d_in)
    fe.check(fe.lib().sd_undistort_points_device(C.c_void_p(d_in.data_ptr()), len(pts), fe._p(K), fe._p(D), C.c_void_p(d_out.data_ptr()), None))
    torch.cuda.synchronize()
    exp = orc.undistort_points(pts, K, D)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    assert np.abs(exp - pts).max() > 5.0            # the distortion is not a no-op
    b = np.zeros(4, np.float32)
    fe.check(fe.lib().sd_image_bounds(640, 480, fe._p(K), fe._p(D), fe._p(b)))
    assert np.array_equal(b, orc.image_bounds(640, 480, K, D)) and b[0] > 5 and b[1] < 635
    Z = np.zeros(5, np.float32)
    fe.check(fe.lib().sd_image_bounds(640, 480, fe._p(K), fe._p(Z), fe._p(b)))
    assert b.tolist() == [0.0, 640.0, 0.0, 480.0]


def test_undistort_keypoints_of_a_batch(gpu, fe, orc, synth):
    import ctypes as C
    import torch
    cfg = synth.TUM3
    img = synth.random_image(cfg["width"], cfg["height"], 5)
    ex = fe.ORBextractor(1000, 1.2, 8, 20, 7)
    b = fe.Batch(ex, cfg["width"], cfg["height"], 1)
    b.extract_host(img[None])
    kp, _, _ = b.download(0)
    K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)
    D = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    d_un = torch.zeros(b.cap * 28, dtype=torch.uint8, device="cuda")
    for dist in (D, np.zeros(5, np.float32)):
        fe.check(fe.lib().sd_batch_undistort_keypoints(b.h, 1, fe._p(K), fe._p(dist), C.c_void_p(d_un.data_ptr()), None))
        b.sync()
        un = d_un.cpu().numpy().view(fe.KP_DTYPE)[:len(kp)]
        exp = kp.copy()
        if dist[0] != 0:
            u = orc.undistort_points(np.stack([kp["x"], kp["y"]], 1), K, dist)
            exp["x"] = u[:, 0]; exp["y"] = u[:, 1]
        assert un.tobytes() == exp.tobytes()
    b.close()


@pytest.mark.parametrize("rgb", [True, False])
def test_extract_color_equals_cvt_then_extract(gpu, fe, orc, synth, rgb):
    """sd_batch_extract_color_device (cvtColor inside the level-0 copy) == sd_cvt_gray_device + sd_batch_extract_device:
    level 0 of the pyramid, keypoints and descriptors, bit for bit; and the gray values are the oracle's."""
    import torch
    cfg = synth.KITTI03_RGBD
    W, H = cfg["width"], cfg["height"]
    frames = [synth.rgbd_frame(3, t, cfg)[0] for t in range(2)]
    rng = np.random.default_rng(5)
    src = np.stack(frames).copy()
    src[..., 1] = np.roll(src[..., 1], 3, axis=2)            # make the three channels differ
    src[..., 2] = (src[..., 2].astype(np.int32) * 3 // 4 + rng.integers(0, 30, src[..., 2].shape)).astype(np.uint8)
    d_src = torch.from_numpy(src).cuda()
    d_gray = torch.empty((2, H, W), dtype=torch.uint8, device="cuda")
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    b1 = fe.Batch(ex, W, H, 2); b2 = fe.Batch(ex, W, H, 2)
    try:
        fe.cvt_gray_device(d_src.data_ptr(), W, H, W * 3, W * H * 3, 3, int(rgb), d_gray.data_ptr(), W, W * H, 2)
        torch.cuda.synchronize()
        b1.extract_device(d_gray.data_ptr(), W, W * H, 2); b1.sync()
        b2.extract_color_device(d_src.data_ptr(), W * 3, W * H * 3, 2, rgb); b2.sync()
        for i in range(2):
            p1, p2 = b1.pyramid(i, 0), b2.pyramid(i, 0)
            assert np.array_equal(p1, p2)
            assert np.array_equal(p2[19:-19, 19:-19], orc.cvt_gray(src[i], rgb))
            k1, d1, l1 = b1.download(i); k2, d2, l2 = b2.download(i)
            assert len(k1) > 1500 and np.array_equal(l1, l2)
            assert k1.tobytes() == k2.tobytes() and np.array_equal(d1, d2)
    finally:
        b1.close(); b2.close()
