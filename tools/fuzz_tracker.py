"""Randomised sweep of the Frame-level boundary (sd_tracker_track) against the frame-level oracle: random lanes (sensor fixed per
run), random box sets per frame (none / empty list / overlapping / partly outside / boxes on flat regions that get erased), jittered
and occasionally jumping time stamps (so that Track_new's loop sees 0, 1 or several candidates), blank and nearly featureless
frames (N == 0, TrackHomo failures), scene cuts; with `state` (round 3) every draw also hands over a caller's SLAM state -- a random pose
prior per frame (small yaw / pitch / translation steps), random mState bits (not initialised, lost), MapPoints committed after random frames
(moved, culled, marked as observed) -- and many-box frames (up to 40 boxes, above the old 32-box tables).  Every frame of every lane is compared bit for bit (tests/test_gpu_pipeline.py's
checker).  usage: fuzz_tracker.py <draws> <seed> [stereo|rgbd] [state]   -> exit code 0 when all draws agree."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402


def _small_pose(rng, prev):
    a, b = rng.normal(0, 0.004), rng.normal(0, 0.002)
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]); Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    M = np.eye(4); M[:3, :3] = Ry @ Rx; M[:3, 3] = rng.normal(0, 0.03, 3)
    return (M @ prev.astype(np.float64)).astype(np.float32)


def add_state(rng, lane, T, seed):
    """The hooks of tests/test_gpu_pipeline._run_chain for a caller that owns SLAM state."""
    poses = [np.eye(4, dtype=np.float32)]
    for _ in range(1, T):
        poses.append(_small_pose(rng, poses[-1]))
    init_frames = int(rng.integers(1, 3))
    lost = int(rng.integers(3, T)) if rng.random() < 0.4 else -1
    states = [0 if t < init_frames else (1 if (t == lost or t == init_frames) else 3) for t in range(T)]
    commit = set(int(t) for t in range(T) if rng.random() < 0.5)

    def mappoints(t, F):
        if t not in commit or F.N == 0:
            return None
        r = np.random.default_rng(seed * 7919 + t)
        xw = F.xw.copy(); fl = F.mp_flags.copy()
        xw += r.normal(0, 0.02, xw.shape).astype(np.float32)
        fl[r.random(len(fl)) < 0.2] = 0
        fl[(r.random(len(fl)) < 0.3) & (fl != 0)] |= 2
        return xw, fl
    lane.update(pose=lambda t: poses[t], state=lambda t: states[t], mappoints=mappoints)
    return lane


def draw_lane(rng, synth, cfg, stereo, T):
    W, H = cfg["width"], cfg["height"]
    seq = int(rng.integers(100, 400))
    fps = cfg["fps"]
    stamps, t = [], 0.0
    for k in range(T):
        stamps.append(t)
        t += (1.0 / fps) * float(rng.choice([1.0, 1.0, 1.0, 0.5, 1.5, 3.0]))
    cut = int(rng.integers(2, T)) if rng.random() < 0.3 else T + 1          # scene cut: frames from another sequence afterwards
    blank = int(rng.integers(1, T)) if rng.random() < 0.3 else -1           # one blank frame (N == 0)
    mode = rng.choice(["boxes", "boxes", "boxes", "mixed", "none"])

    def frame(k):
        s = seq if k < cut else seq + 500
        if stereo:
            a, b, _ = synth.stereo_frame_dyn(s, k, cfg)
        else:
            a, b, _ = synth.rgbd_frame_dyn(s, k, cfg)
        if k == blank:
            a = np.full_like(a, 90)
            if stereo:
                b = np.full_like(b, 90)
        return a, b

    def boxes(k):
        if mode == "none" or (mode == "mixed" and rng.random() < 0.4):
            return None
        r = np.random.default_rng(seq * 131 + k)
        base = synth.rows_to_rects(synth.boxes_for_frame(seq if k < cut else seq + 500, k, cfg))
        out = [base[j] for j in range(len(base)) if r.random() < 0.85]
        for _ in range(int(r.integers(0, 3))):                                 # random extra boxes, some partly outside
            w, h = r.uniform(20, 260), r.uniform(20, 160)
            out.append(np.array([r.uniform(-40, W - 20), r.uniform(-30, H - 20), w, h]))
        if r.random() < 0.3 and len(out):                                     # an overlapping copy of the first box
            out.append(out[0] + np.array([15.0, 8.0, 0.0, 0.0]))
        if r.random() < 0.15:                                                 # a crowded frame: more boxes than the 32-entry tables of round 2 held
            for _ in range(int(r.integers(20, 27))):          # <= 32 detector boxes, so that two crowded frames in a row (<= 64 objects after re-injection) still fit
                w, h = r.uniform(15, 120), r.uniform(15, 90)
                out.append(np.array([r.uniform(0, W - 20), r.uniform(0, H - 20), w, h]))
        if r.random() < 0.2:
            out = []
        return np.array(out, np.float64).reshape(-1, 4)
    cache_b = {}
    return dict(frames=frame, boxes=lambda k: cache_b.setdefault(k, boxes(k)), stamps=stamps)


def run(draws, seed, kind="stereo", lanes=3, T=8, verbose=False, state=False):
    import test_gpu_pipeline as tp
    pkg = graft.load_package(); orc = graft.load_oracle()
    fe, synth = pkg.frontend, pkg.synth
    stereo = kind == "stereo"
    cfg = synth.KITTI_STEREO if stereo else synth.TUM3
    bad = 0
    for d in range(draws):
        rng = np.random.default_rng(seed * 1000 + d)
        L = [draw_lane(rng, synth, cfg, stereo, T) for _ in range(lanes)]
        if state:
            L = [add_state(rng, ln, T, seed * 1000 + d * 10 + k) for k, ln in enumerate(L)]
        try:
            st = tp._run_chain(fe, orc, synth, cfg, fe.SENSOR_STEREO if stereo else fe.SENSOR_RGBD, L, T, channels=1 if stereo else 3)
            if verbose:
                print("draw %d ok: %r" % (d, {k: v for k, v in st.items() if k != "refs"}))
        except AssertionError as e:
            bad += 1
            print("draw %d (seed %d) FAILED: %s" % (d, seed, str(e)[:300]))
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    kind = sys.argv[3] if len(sys.argv) > 3 else "stereo"
    sys.exit(1 if run(n, sd, kind, verbose=True, state=len(sys.argv) > 4 and sys.argv[4] == "state") else 0)
