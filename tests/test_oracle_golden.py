"""The oracle against its committed regression fixtures (tests/golden/, made by tools/gen_golden.py).
These fixtures are self-generated: the reference has none for this path (parity unpinned)."""
import os
import zlib

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def images(synth):
    L, R, _ = synth.stereo_frame(seq=7, t=0)
    return {"extract_kitti_left": L, "extract_kitti_right": R, "extract_tum_640x480": synth.random_image(640, 480, 5, "texture")}


@pytest.mark.parametrize("name", ["extract_kitti_left", "extract_kitti_right", "extract_tum_640x480"])
def test_extract_golden(orc, synth, name):
    g = load(name)
    img = images(synth)[name]
    assert crc(img) == int(g["input_crc"]), "the seeded generator drifted: regenerate fixtures deliberately, not silently"
    nf, ini, mn = [int(v) for v in g["params"]]
    ex = orc.Extractor(nf, 1.2, 8, ini, mn)
    kp, desc = ex(img)
    assert np.array_equal(kp.view(np.uint8).reshape(len(kp), 28), g["kp"])
    assert np.array_equal(desc, g["desc"])
    assert np.array_equal(ex.per_level, g["per_level"]) and np.array_equal(ex.cand_per_level, g["cand_per_level"])
    assert [crc(ex.pyramid(l)) for l in range(8)] == g["pyr_crc"].tolist()
    assert [crc(ex.blurred(l)) for l in range(8)] == g["blur_crc"].tolist()
    # structural properties every ORB-SLAM2 extraction has
    assert nf <= len(kp) <= nf + 2 * 8
    assert (np.diff(kp["octave"]) >= 0).all()                  # concatenated level by level
    assert ((kp["angle"] >= 0) & (kp["angle"] < 360)).all()
    assert (kp["response"] >= mn).all()
    assert kp["class_id"].tolist() == [-1] * len(kp)


def test_stereo_and_projection_golden(orc, synth, fe):
    cfg = synth.KITTI_STEREO
    L, R, _ = synth.stereo_frame(seq=7, t=0)
    eL = orc.Extractor(2000, 1.2, 8, 12, 7); eR = orc.Extractor(2000, 1.2, 8, 12, 7)
    kL, dL = eL(L); kR, dR = eR(R)
    ur, dep, sad, nm = orc.stereo_matches(eL, eR, kL, dL, kR, dR, cfg["bf"], cfg["fx"])
    g = load("stereo_kitti")
    assert nm == int(g["nmatched"]) and np.array_equal(sad, g["sad"])
    assert np.array_equal(ur.view(np.uint32), g["uright"].view(np.uint32))
    assert np.array_equal(dep.view(np.uint32), g["depth"].view(np.uint32))
    ok = ur >= 0
    assert ok.sum() > 500
    assert (ur[ok] <= kL["x"][ok]).all() and (dep[ok] > 0).all()          # disparity >= 0 (Frame.cc:1017-1024)
    # synthetic disparities are integers in [4, 64]: recovered disparity must land there
    disp = kL["x"][ok] - ur[ok]
    assert (disp > 3).mean() > 0.95 and (disp < 66).mean() > 0.95
    L1, R1, _ = synth.stereo_frame(seq=7, t=1)
    e1 = orc.Extractor(2000, 1.2, 8, 12, 7); e1r = orc.Extractor(2000, 1.2, 8, 12, 7)
    k1, d1 = e1(L1); k1r, d1r = e1r(R1)
    ur1, dep1, _, _ = orc.stereo_matches(e1, e1r, k1, d1, k1r, d1r, cfg["bf"], cfg["fx"])
    cam10 = fe.camera_array(fe.make_camera(cfg))
    I = np.eye(4, dtype=np.float32)
    xw, valid = orc.unproject(kL, dep, cam10, I)
    m, pairs, nmatch = orc.search_by_projection(k1, d1, ur1, kL, dL, xw, valid, I, I, cam10, eL.scale, 7.0)
    g = load("projection_kitti_t0_t1")
    assert nmatch == int(g["nmatches"]) and np.array_equal(m, g["match"]) and np.array_equal(pairs, g["pairs"])
    assert np.array_equal(xw.view(np.uint32), g["xw"].view(np.uint32)) and np.array_equal(valid, g["valid"])
    assert np.array_equal(orc.grid_cells(k1, cam10), g["grid"])
    assert nmatch > 300


def test_micro_golden(orc):
    g = load("micro")
    ham = [orc.descriptor_distance(a, b) for a, b in zip(g["ham_a"], g["ham_b"])]
    assert ham == g["ham"].tolist()
    at = np.array([orc.fast_atan2(y, x) for y, x in g["atan_yx"]], np.float32)
    assert np.array_equal(at.view(np.uint32), g["atan"].view(np.uint32))
