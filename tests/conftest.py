import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def fe(pkg):
    if not os.path.exists(pkg.frontend.LIB_PATH):
        graft.build()
    return pkg.frontend


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def orc():
    return graft.load_oracle()


@pytest.fixture(scope="session")
def gpu(fe):
    """Fail (not skip) when a -m gpu test runs without a usable device: no silent fallback."""
    n = fe.device_count()
    assert n >= 1, "no HIP device visible: GPU tests must run on the GPU box"
    return n


def assert_kp_equal(a, b, what=""):
    assert len(a) == len(b), "%s: %d vs %d keypoints" % (what, len(a), len(b))
    for f in a.dtype.names:
        if not np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)):
            bad = np.nonzero(a[f].view(np.uint32) != b[f].view(np.uint32))[0]
            raise AssertionError("%s: field %s differs at %d positions, first %d: %r vs %r"
                                 % (what, f, len(bad), bad[0], a[bad[0]], b[bad[0]]))
