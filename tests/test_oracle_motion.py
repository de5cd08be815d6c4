"""CPU tests of the TrackHomo model-fit oracle (oracle/motion_oracle.inc, spec Q13) on synthetic correspondences."""
import numpy as np


def _scene(rng, N, outlier_frac, noise):
    p1 = np.stack([rng.uniform(0, 1241, N), rng.uniform(0, 376, N)], 1).astype(np.float32)
    Ht = np.array([[1.02, 0.01, 6.0], [-0.005, 1.02, -3.0], [1e-5, -2e-5, 1.0]])
    q = (Ht @ np.c_[p1, np.ones(N)].T).T
    p2 = (q[:, :2] / q[:, 2:] + rng.normal(0, noise, (N, 2))).astype(np.float32)
    out = rng.random(N) < outlier_frac
    p2[out] = np.stack([rng.uniform(0, 1241, out.sum()), rng.uniform(0, 376, out.sum())], 1)
    return p1, p2, out, Ht


def test_homography_recovered_with_outliers(orc):
    rng = np.random.default_rng(3)
    p1, p2, out, Ht = _scene(rng, 900, 0.3, 0.4)
    r = orc.estimate_motion(p1, p2)
    assert r["flag"] in (1, 2) and r["n_h"] >= 0.97 * (~out).sum()
    assert r["mask_h"][~out].mean() > 0.97 and r["mask_h"][out].mean() < 0.05
    assert np.abs(r["H"] - Ht).max() < 0.1 and abs(r["H"][2, 2] - 1) < 1e-12
    # F from a planar scene is degenerate but must still be rank 2 and fit its own inliers
    assert abs(np.linalg.det(r["F"])) < 1e-12 * np.abs(r["F"]).max() ** 3 + 1e-18
    assert r["n_f"] == int(r["mask_f"].sum()) and r["n_h"] == int(r["mask_h"].sum())


def test_general_motion_prefers_fundamental(orc):
    """Points at very different depths under a sideways translation: no single homography explains them, F does."""
    rng = np.random.default_rng(4)
    N = 700
    X = np.stack([rng.uniform(-8, 8, N), rng.uniform(-2, 2, N), rng.uniform(4, 40, N)], 1)
    K = np.array([[707.0, 0, 601.9], [0, 707.0, 183.1], [0, 0, 1]])
    p1 = (K @ X.T).T; p1 = p1[:, :2] / p1[:, 2:]
    X2 = X + np.array([0.9, 0.05, -0.3])
    p2 = (K @ X2.T).T; p2 = p2[:, :2] / p2[:, 2:]
    p2 += rng.normal(0, 0.3, p2.shape)
    r = orc.estimate_motion(p1.astype(np.float32), p2.astype(np.float32))
    assert r["flag"] == 2 and r["n_f"] > 0.9 * N and r["n_h"] < r["n_f"]
    x1 = np.c_[p1, np.ones(N)]; x2 = np.c_[p2, np.ones(N)]
    l = (r["F"] @ x1.T).T
    d = np.abs(np.sum(l * x2, 1)) / np.hypot(l[:, 0], l[:, 1])
    assert np.median(d) < 1.0


def test_degenerate_inputs(orc):
    r = orc.estimate_motion(np.zeros((5, 2), np.float32), np.zeros((5, 2), np.float32))
    assert r["flag"] == 0 and r["n_h"] == 0 and r["n_f"] == 0                       # fewer than 8 pairs
    p = np.tile(np.array([[10.0, 20.0]], np.float32), (50, 1))
    r = orc.estimate_motion(p, p)
    assert r["flag"] == 0                                                          # zero spread: normalisation impossible
