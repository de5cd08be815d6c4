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
