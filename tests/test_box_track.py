"""Frame::boxTrack through the C ABI (host code) against the oracle — no GPU needed."""
import numpy as np


def _rand_boxes(rng, n, W=1241, H=376):
    w = rng.uniform(40, 250, n); h = rng.uniform(30, 170, n)
    x = rng.uniform(0, W - 60, n); y = rng.uniform(0, H - 40, n)
    return np.stack([x, y, w, h], 1)


def test_box_track_matches_oracle(fe, orc):
    rng = np.random.default_rng(77)
    for trial in range(200):
        n_last = int(rng.integers(0, 7)); n_cur = int(rng.integers(0, 7))
        last = _rand_boxes(rng, n_last)
        # current boxes: some are the last ones moved a little (good IoU), some new
        cur = _rand_boxes(rng, n_cur)
        for k in range(min(n_cur, n_last)):
            if rng.random() < 0.6:
                cur[k] = last[rng.integers(0, n_last)] + np.array([rng.uniform(-8, 8), rng.uniform(-4, 4), 0, 0])
        last_idx = rng.permutation(20)[:n_last].astype(np.int32)
        last_omit = (rng.random(n_last) < 0.25).astype(np.uint8)
        last_vel = rng.uniform(-6, 6, (n_last, 2))
        a = fe.box_track(cur, last, last_idx, last_omit, last_vel, 1241, 376)
        b = orc.box_track(cur, last, last_idx, last_omit, last_vel, 1241, 376)
        for x, y in zip(a, b):
            assert x.shape == y.shape and np.array_equal(x, y), "trial %d" % trial


def test_box_track_first_frame_and_reinjection(fe):
    boxes = np.array([[10., 10, 50, 50], [200, 100, 80, 60]])
    b, idx, om, vel = fe.box_track(boxes, np.zeros((0, 4)), [], [], np.zeros((0, 2)), 640, 480)
    assert idx.tolist() == [0, 1] and om.tolist() == [0, 0] and not vel.any()        # re-initialised: ids 0..n-1
    # next frame: box 0 moved, box 1 vanished -> it is re-injected once with omit = 1 at its predicted position
    cur = np.array([[14., 11, 50, 50]])
    b2, idx2, om2, vel2 = fe.box_track(cur, b, idx, om, np.array([[0., 0], [3., 1]]), 640, 480)
    assert idx2.tolist() == [0, 1] and om2.tolist() == [0, 1]
    assert np.allclose(vel2[0], [4, 1]) and np.allclose(b2[1], [203, 101, 80, 60])
    # a frame later the omitted box is not injected again
    b3, idx3, om3, _ = fe.box_track(cur, b2, idx2, om2, vel2, 640, 480)
    assert idx3.tolist() == [0]
