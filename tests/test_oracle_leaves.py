"""The OpenCV leaf restatements (oracle/cv_leaves.h) against independent formulations and hand-checkable cases."""
import math

import numpy as np
import pytest

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
        (-3, 1), (-2, 2), (-1, 3)]


def brute_fast(img, thr):
    """FAST-9/16 straight from its definition: corner iff >= 9 contiguous ring pixels all darker than v-t or all
    brighter than v+t; score = the largest t for which that still holds; 3x3 strict-maximum NMS; row-major output."""
    h, w = img.shape
    I = img.astype(np.int32)

    def is_corner(y, x, t):
        v = I[y, x]
        d = [v - I[y + dy, x + dx] for dx, dy in RING]
        for sign in (1, -1):
            flags = [sign * e > t for e in d] * 2
            run = 0
            for f in flags:
                run = run + 1 if f else 0
                if run >= 9:
                    return True
        return False

    score = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if is_corner(y, x, thr):
                t = thr
                while t < 255 and is_corner(y, x, t + 1):
                    t += 1
                score[y, x] = t
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s == 0:
                continue
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, s))
    return np.array(out, np.int32).reshape(-1, 3)


def test_fast_against_definition(orc):
    rng = np.random.default_rng(3)
    for seed, thr in ((1, 20), (2, 7), (3, 12)):
        img = rng.integers(0, 256, (34, 41), dtype=np.uint8)
        img[8:20, 10:25] = 200          # structure: rectangle corners + noise
        img[22:30, 5:12] = 30
        got = orc.fast(img, thr)
        ref = brute_fast(img, thr)
        assert np.array_equal(got, ref), "thr %d: %d vs %d corners" % (thr, len(got), len(ref))
        assert len(ref) > 5


def test_fast_single_bright_dot(orc):
    img = np.full((21, 21), 50, np.uint8)
    img[10, 10] = 150                    # all 16 ring pixels darker by 100 -> corner, score 99
    got = orc.fast(img, 20)
    assert got.tolist() == [[10, 10, 99]]
    assert len(orc.fast(np.full((21, 21), 50, np.uint8), 7)) == 0


def test_resize_properties(orc):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (60, 73), dtype=np.uint8)
    assert np.array_equal(orc.resize_linear(img, 73, 60), img)                  # identity size = copy
    const = np.full((50, 61), 93, np.uint8)
    assert (orc.resize_linear(const, 51, 42) == 93).all()                       # constants survive the 11-bit lerp
    # exact 2:1 along x with centred samples: dst = (a+b)/2 rounded (weights 1024/1024)
    a = np.zeros((4, 8), np.uint8); a[:, ::2] = 10; a[:, 1::2] = 31
    assert (orc.resize_linear(a, 4, 4) == 21).all()                             # (10*1024+31*1024) fixed point -> 20.5 -> 21


def test_gaussian_properties(orc):
    const = np.full((40, 44), 177, np.uint8)
    assert (orc.gaussian7(const) == 177).all()
    imp = np.zeros((21, 21), np.uint8); imp[10, 10] = 255
    taps = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    ref = (255 * np.outer(taps, taps) + 0x8000) >> 16
    assert np.array_equal(orc.gaussian7(imp)[7:14, 7:14], ref.astype(np.uint8))
    # REFLECT_101 at the border: an impulse at (0,0) folds its left/top taps back inside
    imp = np.zeros((21, 21), np.uint8); imp[0, 0] = 255
    t1 = np.array([56, 34 + 34, 48 + 48, 18 + 18], np.int64)[:4]   # k0, k1+k-1 ... for positions 0..3: only col 0 source
    out = orc.gaussian7(imp)
    col = np.array([56, 34, 48, 18], np.int64)    # response at x = 0,1,2,3 to a source at 0 is tap[3 + x] (+ mirrored tap at -x for x>0? no: source unique)
    # response(x) = sum_k tap[k] * [reflect(x + k - 3) == 0]
    resp = np.zeros(4, np.int64)
    for x in range(4):
        for k in range(7):
            p = x + k - 3
            p = -p if p < 0 else p
            if p == 0:
                resp[x] += taps[k]
    ref = (255 * np.outer(resp, resp) + 0x8000) >> 16
    assert np.array_equal(out[:4, :4], ref.astype(np.uint8))


def test_fast_atan2(orc):
    rng = np.random.default_rng(1)
    worst = 0.0
    for _ in range(2000):
        y, x = rng.normal(size=2) * 100
        a = orc.fast_atan2(y, x)
        t = math.degrees(math.atan2(np.float32(y), np.float32(x))) % 360.0
        d = abs(a - t); d = min(d, 360 - d)
        worst = max(worst, d)
        assert 0.0 <= a <= 360.0
    assert worst < 0.3                                    # OpenCV documents ~0.3 degree accuracy
    assert orc.fast_atan2(0, 0) == 0.0 and orc.fast_atan2(0, 1) == 0.0
    assert abs(orc.fast_atan2(1, 0) - 90) < 0.05 and abs(orc.fast_atan2(0, -1) - 180) < 0.05 and abs(orc.fast_atan2(-1, 0) - 270) < 0.05


def test_cvt_gray(orc):
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    rgb = orc.cvt_gray(px, 1)[0].tolist()
    bgr = orc.cvt_gray(px, 0)[0].tolist()
    f = lambda r, g, b: (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14
    assert rgb == [255, f(255, 0, 0), f(0, 255, 0), f(0, 0, 255), f(10, 20, 30)] == [255, 76, 150, 29, 18]
    assert bgr == [255, f(0, 0, 255), f(0, 255, 0), f(255, 0, 0), f(30, 20, 10)]


def test_descriptor_distance(orc, fe):
    rng = np.random.default_rng(2)
    for _ in range(50):
        a = rng.integers(0, 256, 32, dtype=np.uint8); b = rng.integers(0, 256, 32, dtype=np.uint8)
        ref = int(np.unpackbits(a ^ b).sum())
        assert orc.descriptor_distance(a, b) == ref == fe.DescriptorDistance(a, b)
    z = np.zeros(32, np.uint8)
    assert orc.descriptor_distance(z, z) == 0 and orc.descriptor_distance(z, ~z) == 256


def test_steering_cos_sin_equals_libm():
    """The kernels' f64 cos / sin of the steering angle (slam-dynamic_amd/csrc/sd_trig.h: IEEE fma / multiply / rint only, so host ==
    device) gives the f32 values of the C library's cos() / sin() that the oracle uses: every 61st f32 angle in degrees [0.001, 360]
    here (2.5 M values); `cos_sin_check 1` checks all 154 M (no difference when written)."""
    import os, shutil, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++"):
        pytest.skip("no g++")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "cos_sin_check")
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I" + os.path.join(root, "slam-dynamic_amd", "csrc"), "-o", exe,
                               os.path.join(root, "tests", "cpp", "cos_sin_check.cpp")])
        out = subprocess.run([exe, "61"], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout
