"""The algebra behind the detector's SD_YOLO_F32W mode (k_yolo32w.h), checked on the CPU in float64: with the matrices the kernels use,
Y = A^T [ (G g G^T) .* (B^T d B) ] A equals the direct 3 x 3 correlation of a 4 x 4 patch, and the fold coefficients of k_wino_gemm_f32
(Y[a][b] += s_a(xi / 4) * s_b(xi % 4) * M_xi) are the entries of A^T (x) A^T."""
import numpy as np

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def test_f2x2_3x3_equals_direct_correlation():
    rng = np.random.default_rng(0)
    for _ in range(20):
        g = rng.standard_normal((3, 3)); d = rng.standard_normal((4, 4))
        direct = np.array([[np.sum(g * d[a:a + 3, b:b + 3]) for b in range(2)] for a in range(2)])
        wino = At @ ((G @ g @ G.T) * (Bt @ d @ Bt.T)) @ At.T
        assert np.allclose(wino, direct, rtol=1e-12, atol=1e-12)


def test_fold_coefficients_are_the_output_transform():
    # the kernel's fold: row coefficients {xy < 3, (0, 1, -1, -1)[xy]}, column coefficients the same in xx; all 0 / +-1
    s = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    assert np.array_equal(s, At)
    rng = np.random.default_rng(1)
    M = rng.standard_normal((4, 4))
    Y = np.zeros((2, 2))
    for xi in range(16):
        for a in range(2):
            for b in range(2):
                Y[a, b] += s[a, xi >> 2] * s[b, xi & 3] * M[xi >> 2, xi & 3]
    assert np.allclose(Y, At @ M @ At.T, rtol=1e-13, atol=1e-13)
    assert sum(1 for xi in range(16) for a in range(2) for b in range(2) if s[a, xi >> 2] * s[b, xi & 3] != 0) == 36      # 9 of 16 per output


def test_input_transform_as_the_kernel_writes_it():
    # k_wino_input: rows first (e = B^T d), then columns (V = e B), with the four-term forms of the kernel
    rng = np.random.default_rng(2)
    d = rng.standard_normal((4, 4))
    e = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])
    V = np.stack([e[:, 0] - e[:, 2], e[:, 1] + e[:, 2], e[:, 2] - e[:, 1], e[:, 1] - e[:, 3]], axis=1)
    assert np.allclose(V, Bt @ d @ Bt.T, rtol=1e-13, atol=1e-13)


def test_weight_transform_as_the_host_writes_it():
    # sd_yolo_load_darknet_weights: t = G g, then U = t G^T, with the halves applied after the sums
    rng = np.random.default_rng(3)
    g = rng.standard_normal((3, 3))
    t = np.stack([g[0], 0.5 * ((g[0] + g[1]) + g[2]), 0.5 * ((g[0] - g[1]) + g[2]), g[2]])
    U = np.stack([t[:, 0], 0.5 * ((t[:, 0] + t[:, 1]) + t[:, 2]), 0.5 * ((t[:, 0] - t[:, 1]) + t[:, 2]), t[:, 2]], axis=1)
    assert np.allclose(U, G @ g @ G.T, rtol=1e-13, atol=1e-13)
