"""host/Trajectory.h (System::SaveTrajectoryTUM / SaveTrajectoryKITTI, src/System.cc:434-565, for caller-supplied poses) against the numpy
restatement, text for text.  Pure host code: runs without a GPU."""
import importlib.util
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle():
    spec = importlib.util.spec_from_file_location("sd_trajectory_oracle", os.path.join(ROOT, "oracle", "trajectory_oracle.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    R = np.eye(3)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    R[i, i] = c; R[j, j] = c; R[i, j] = -s; R[j, i] = s
    return R


def test_trajectory_text_matches_oracle(tmp_path):
    T = _oracle()
    rng = np.random.default_rng(5)
    poses, stamps, lost = [], [], []
    for k in range(40):
        # every branch of Eigen's matrix -> quaternion: small rotations (trace > 0) and half turns about each axis (largest diagonal element)
        if k < 20:
            R = _rot(0, rng.normal(0, 0.3)) @ _rot(1, rng.normal(0, 0.3)) @ _rot(2, rng.normal(0, 0.3))
        else:
            R = _rot(k % 3, np.pi - rng.uniform(0, 0.2)) @ _rot((k + 1) % 3, rng.normal(0, 0.05))
        M = np.eye(4, dtype=np.float32)
        M[:3, :3] = R.astype(np.float32); M[:3, 3] = rng.normal(0, 20, 3).astype(np.float32)
        poses.append(M); stamps.append(1317384506.0 + 0.1 * k + rng.uniform(0, 1e-3)); lost.append(k in (7, 8, 31))
    blob = b"".join(M.tobytes() + np.array([ts], np.float64).tobytes() + np.array([int(l)], np.int32).tobytes() for M, ts, l in zip(poses, stamps, lost))
    (tmp_path / "in.bin").write_bytes(blob)
    exe = str(tmp_path / "traj")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-ffp-contract=off", "-I" + os.path.join(ROOT, "slam-dynamic_amd", "host"),
                           os.path.join(ROOT, "tests/cpp/trajectory_main.cpp"), "-o", exe])
    subprocess.check_call([exe, str(tmp_path / "in.bin"), str(len(poses)), str(tmp_path / "tum.txt"), str(tmp_path / "kitti.txt")])
    tum = (tmp_path / "tum.txt").read_text(); kitti = (tmp_path / "kitti.txt").read_text()
    assert tum == T.tum_text(poses, stamps, lost)
    assert kitti == T.kitti_text(poses)
    assert len(tum.splitlines()) == 37 and len(kitti.splitlines()) == 40           # TUM skips the lost frames, KITTI does not
    q = np.array([[float(v) for v in line.split()[4:]] for line in tum.splitlines()])
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-5) and (q[:, 3] < 0.2).any() and (q[:, 3] > 0.8).any()
