"""Known-answer material the reference source holds for this path: the rBRIEF pattern,
thresholds, and the tables the ORBextractor constructor derives.  Pins the oracle AND the C ABI host code."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/ORBextractor.cc"


def _inc(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return np.array([int(x) for x in re.findall(r"-?\d+", txt)], np.int32)


def test_pattern_tables_equal_fixture():
    gold = np.loadtxt(os.path.join(ROOT, "tests/golden/orb_pattern_i8.txt"), dtype=np.int32, comments="#").reshape(-1)
    assert gold.shape == (1024,) and np.abs(gold).max() == 13
    assert np.array_equal(_inc(os.path.join(ROOT, "oracle/orb_pattern.inc")), gold)
    assert np.array_equal(_inc(os.path.join(ROOT, "slam-dynamic_amd/csrc/orb_pattern.inc")), gold)


def test_pattern_equals_reference_source_when_present():
    """In this container the reference tree is mounted: the fixture is checked against bit_pattern_31_ itself
    (src/ORBextractor.cc:150-408).  On the GPU box (no /root/reference) the committed fixture stands in."""
    if not os.path.exists(REF):
        return
    src = open(REF).read()
    i = src.index("bit_pattern_31_[256*4]")
    body = src[i:src.index("};", i)]
    body = re.sub(r"/\*.*?\*/", "", body[body.index("{") + 1:], flags=re.S)
    ref = np.array([int(x) for x in re.findall(r"-?\d+", body)], np.int32)
    gold = np.loadtxt(os.path.join(ROOT, "tests/golden/orb_pattern_i8.txt"), dtype=np.int32, comments="#").reshape(-1)
    assert np.array_equal(ref, gold)
    # thresholds the oracle hard-codes (ORBextractor.cc:72-74, ORBmatcher.cc:37-39)
    assert "const int PATCH_SIZE = 31;" in src and "const int HALF_PATCH_SIZE = 15;" in src and "const int EDGE_THRESHOLD = 19;" in src
    m = open("/root/reference/src/ORBmatcher.cc").read()
    assert "TH_HIGH = 100" in m and "TH_LOW = 50" in m and "HISTO_LENGTH = 30" in m


# SURVEY.md section 8a row 2 / Appendix B (computed from ORBextractor.cc:410-446,456-470,1111-1112)
QUOTAS = {(2000, 8): [434, 362, 302, 251, 209, 175, 145, 122], (1000, 8): [217, 181, 151, 126, 105, 87, 73, 60],
          (1500, 8): [326, 271, 226, 189, 157, 131, 109, 91], (3000, 8): [652, 543, 452, 377, 314, 262, 218, 182]}
UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
SIZES_KITTI = [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
SIZES_TUM = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]


def test_extractor_tables_oracle(orc):
    for (nf, nl), q in QUOTAS.items():
        o = orc.Extractor(nf, 1.2, nl, 20, 7)
        assert list(o.quota) == q
        assert list(o.umax) == UMAX
        assert o.scale[0] == 1.0 and abs(o.scale[7] - 1.2 ** 7) < 1e-5
        assert np.array_equal(o.inv_scale, (np.float32(1.0) / o.scale).astype(np.float32))
        assert np.array_equal(o.sigma2, o.scale * o.scale)


def test_extractor_tables_cabi(fe, orc):
    for (nf, nl), q in QUOTAS.items():
        ex = fe.ORBextractor(nf, 1.2, nl, 20, 7)
        o = orc.Extractor(nf, 1.2, nl, 20, 7)
        assert list(ex.mnFeaturesPerLevel) == q and list(ex.umax) == UMAX
        for a, b in ((ex.mvScaleFactor, o.scale), (ex.mvInvScaleFactor, o.inv_scale), (ex.mvLevelSigma2, o.sigma2),
                     (ex.mvInvLevelSigma2, o.inv_sigma2)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert ex.GetLevels() == nl and abs(ex.GetScaleFactor() - 1.2) < 1e-7
    ex = fe.ORBextractor(2000, 1.2, 8, 20, 7)
    assert [ex.level_size(1241, 376, l) for l in range(8)] == SIZES_KITTI
    assert [ex.level_size(640, 480, l) for l in range(8)] == SIZES_TUM


def test_pyramid_sizes_oracle(orc, synth):
    o = orc.Extractor(500, 1.2, 8, 20, 7)
    o(synth.random_image(640, 480, 1))
    assert [(o.pyramid(l).shape[1] - 38, o.pyramid(l).shape[0] - 38) for l in range(8)] == SIZES_TUM
