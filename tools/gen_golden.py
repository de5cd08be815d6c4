#!/usr/bin/env python3
"""Generate the regression fixtures under tests/golden/ from the CPU oracle.

The reference ships no golden vectors and cannot be run here (OpenCV absent), so these
fixtures pin the ORACLE (and, through the -m gpu tests, the HIP path) against change; they do
not pin either against the real reference: "parity unpinned" (DESIGN.md).  Inputs are
regenerated from the seeded generator (slam-dynamic_amd/synth.py); each fixture stores an
input CRC so a drifting generator is detected.

  python tools/gen_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
orc = graft.load_oracle()
synth, fe = pkg.synth, pkg.frontend
OUT = os.path.join(ROOT, "tests", "golden")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def extract_fixture(name, img, nf, ini, mn):
    ex = orc.Extractor(nf, 1.2, 8, ini, mn)
    kp, desc = ex(img)
    pyr = np.array([crc(ex.pyramid(l)) for l in range(8)], np.uint32)
    blur = np.array([crc(ex.blurred(l)) for l in range(8)], np.uint32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), input_crc=np.uint32(crc(img)), params=np.array([nf, ini, mn]),
                        kp=kp.view(np.uint8).reshape(len(kp), 28), desc=desc, per_level=ex.per_level,
                        cand_per_level=ex.cand_per_level, pyr_crc=pyr, blur_crc=blur)
    return ex, kp, desc


def main():
    cfg = synth.KITTI_STEREO
    L, R, _ = synth.stereo_frame(seq=7, t=0)
    exL, kL, dL = extract_fixture("extract_kitti_left", L, 2000, 12, 7)
    exR, kR, dR = extract_fixture("extract_kitti_right", R, 2000, 12, 7)
    ur, dep, sad, nm = orc.stereo_matches(exL, exR, kL, dL, kR, dR, cfg["bf"], cfg["fx"])
    np.savez_compressed(os.path.join(OUT, "stereo_kitti.npz"), uright=ur, depth=dep, sad=sad, nmatched=np.int32(nm))
    tum = synth.random_image(640, 480, 5, "texture")
    extract_fixture("extract_tum_640x480", tum, 1000, 20, 7)
    # frame-to-frame projection match between t=0 and t=1 of the same sequence
    L1, R1, _ = synth.stereo_frame(seq=7, t=1)
    e1 = orc.Extractor(2000, 1.2, 8, 12, 7); e1r = orc.Extractor(2000, 1.2, 8, 12, 7)
    k1, d1 = e1(L1); k1r, d1r = e1r(R1)
    ur1, dep1, _, _ = orc.stereo_matches(e1, e1r, k1, d1, k1r, d1r, cfg["bf"], cfg["fx"])
    cam10 = fe.camera_array(fe.make_camera(cfg))
    I = np.eye(4, dtype=np.float32)
    xw, valid = orc.unproject(kL, dep, cam10, I)
    m, pairs, nmatch = orc.search_by_projection(k1, d1, ur1, kL, dL, xw, valid, I, I, cam10, exL.scale, 7.0)
    np.savez_compressed(os.path.join(OUT, "projection_kitti_t0_t1.npz"), match=m, pairs=pairs, nmatches=np.int32(nmatch),
                        xw=xw, valid=valid, grid=orc.grid_cells(k1, cam10))
    # micro-vectors: Hamming, fastAtan2, gray
    rng = np.random.default_rng(42)
    a = rng.integers(0, 256, (64, 32), dtype=np.uint8); b = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    ham = np.array([orc.descriptor_distance(a[i], b[i]) for i in range(64)], np.int32)
    yx = rng.normal(size=(256, 2)).astype(np.float32) * 1000
    at = np.array([orc.fast_atan2(y, x) for y, x in yx], np.float32)
    np.savez_compressed(os.path.join(OUT, "micro.npz"), ham_a=a, ham_b=b, ham=ham, atan_yx=yx, atan=at)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
