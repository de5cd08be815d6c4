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


def _py_refqueue():
    """Literal restatement of the q_frame handling (Tracking.cc:620-666, 952-959) for comparison."""
    class Q:
        def __init__(self): self.q = []
        def candidate(self, t, has):
            if not has: return -1
            while self.q and t - self.q[0][0] > np.float32(0.2):
                if not self.q[0][2]:
                    self.q.pop(0); continue
                return self.q[0][1]
            return -1
        def reject(self):
            if len(self.q) <= 1: return False
            self.q.pop(0); return True
        def push(self, t, slot, has, fps):
            ev = -1
            if len(self.q) >= fps * 0.3 and self.q:
                ev = self.q.pop(0)[1]
            self.q.append((t, slot, has)); return ev
    return Q()


def test_reference_queue(fe):
    rng = np.random.default_rng(5)
    for fps in (10, 30):
        q, r = fe.RefQueue(), _py_refqueue()
        t = 0.0
        for k in range(300):
            t += 1.0 / fps
            has = bool(rng.random() < 0.8)
            c1, c2 = q.candidate(t, has), r.candidate(t, has)
            assert c1 == c2
            while c1 >= 0 and rng.random() < 0.3:        # TrackHomo "fails": try the next-oldest frame
                a1, a2 = q.reject(), r.reject()
                assert a1 == a2
                if not a1: break
                c1, c2 = q.candidate(t, has), r.candidate(t, has)
                assert c1 == c2
            if rng.random() < 0.9:                        # mState == OK
                assert q.push(t, k % 16, has, fps) == r.push(t, k % 16, has, fps)
            assert len(q) == len(r.q) <= max(1, int(np.ceil(fps * 0.3)))
    q = fe.RefQueue()
    for k in range(5):
        q.push(0.1 * k, k, True, 10)
    assert len(q) == 3                                     # 0.3 * fps frames at 10 fps
    assert q.candidate(0.55, True) == 2 and q.candidate(0.55, False) == -1
