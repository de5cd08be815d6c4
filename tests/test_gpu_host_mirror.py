"""The C++ class-API mirror (slam-dynamic_amd/host/ORBextractor.h) built with g++ against the C ABI and run on the GPU."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_orbextractor_matches_oracle(gpu, fe, orc, synth, tmp_path):
    exe = str(tmp_path / "host_mirror")
    libdir = os.path.join(ROOT, "slam-dynamic_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "slam-dynamic_amd", "host"), os.path.join(ROOT, "tests/cpp/host_mirror_main.cpp"),
                           "-L" + libdir, "-lsd_frontend", "-Wl,-rpath," + libdir, "-o", exe])
    img = synth.random_image(640, 480, 31)
    raw = tmp_path / "img.raw"; out = tmp_path / "out.bin"
    raw.write_bytes(img.tobytes())
    subprocess.check_call([exe, "640", "480", str(raw), str(out), "1000", "20", "7"])
    blob = out.read_bytes()
    n = int(np.frombuffer(blob, np.int32, 1)[0])
    kp = np.frombuffer(blob, fe.KP_DTYPE, n, 4)
    desc = np.frombuffer(blob, np.uint8, n * 32, 4 + 28 * n).reshape(n, 32)
    o = orc.Extractor(1000, 1.2, 8, 20, 7)
    rk, rd = o(img)
    assert n == len(rk) and kp.tobytes() == rk.tobytes() and np.array_equal(desc, rd)
    off = 4 + 60 * n
    w1, h1 = np.frombuffer(blob, np.int32, 2, off)
    plane = np.frombuffer(blob, np.uint8, (w1 + 38) * (h1 + 38), off + 8).reshape(h1 + 38, w1 + 38)
    assert np.array_equal(plane, o.pyramid(1))
