"""Pins the oracle's BowVector / FeatureVector (oracle/bow_oracle.inc) against the REFERENCE's own code: the two OpenCV-free
translation units Thirdparty/DBoW2/DBoW2/BowVector.cpp and FeatureVector.cpp, compiled unmodified from /root/reference into
oracle/_ref/libdbow2_ref.so (oracle/Makefile target `ref`; built by __graft_entry__.build() where the reference is mounted).
This is the only piece of the hot path's arithmetic the image can build from the reference (everything else needs OpenCV)."""
import ctypes as C
import os

import numpy as np
import pytest

import __graft_entry__ as graft

REF_SO = os.path.join(graft.ROOT, "oracle", "_ref", "libdbow2_ref.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF_SO):
        if not os.path.isdir("/root/reference/Thirdparty/DBoW2/DBoW2"):
            pytest.skip("oracle/_ref is not built and /root/reference is not mounted")
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(graft.ROOT, "oracle"), "-s", "ref"])
    return C.CDLL(REF_SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _ref_bow(ref, word, weight, node, normalize=True):
    n = len(word)
    bw = np.zeros(n, np.uint32); bv = np.zeros(n, np.float64); fn = np.zeros(n, np.uint32); ff = np.zeros(n, np.uint32); nf = C.c_int()
    k = ref.ref_bow_from_features(_p(np.ascontiguousarray(word, np.uint32)), _p(np.ascontiguousarray(weight, np.float64)),
                                  _p(np.ascontiguousarray(node, np.uint32)), n, int(normalize), _p(bw), _p(bv), _p(fn), _p(ff), C.byref(nf))
    return bw[:k], bv[:k], fn[:nf.value], ff[:nf.value]


@pytest.mark.parametrize("L,levelsup,n_desc,seed", [(3, 2, 500, 1), (4, 4, 2000, 2), (6, 4, 2000, 3)])
def test_bow_and_feature_vectors_equal_the_reference_classes(ref, orc, synth, L, levelsup, n_desc, seed):
    voc = synth.vocabulary(k=10, L=L, seed=seed, stop_frac=0.05)
    O = orc.Vocabulary.from_nodes(voc)
    rng = np.random.default_rng(seed)
    # descriptors near vocabulary words (so that words repeat: addWeight's accumulation order matters) + random ones
    leaves = np.nonzero(voc["is_leaf"])[0]
    pick = rng.choice(leaves, n_desc // 2)
    near = voc["desc"][pick].copy()
    flip = rng.integers(0, 256, (len(near), 6))
    for j in range(6):
        near[np.arange(len(near)), flip[:, j] >> 3] ^= (1 << (flip[:, j] & 7)).astype(np.uint8)
    desc = np.concatenate([near, near[: n_desc // 4], rng.integers(0, 256, (n_desc - len(near) - n_desc // 4, 32), dtype=np.uint8)])
    word, weight, node = O.transform(desc, levelsup)
    assert (weight == 0).any() or L == 3, "stop words (weight 0) should occur"
    got = O.compute_bow(desc, levelsup)
    bw, bv, fn, ff = _ref_bow(ref, word, weight, node)
    assert len(bw) < (weight > 0).sum(), "words must repeat"
    assert np.array_equal(got["word"], bw), "BowVector ids"
    assert np.array_equal(got["value"].view(np.uint64), bv.view(np.uint64)), "BowVector values must be BIT-identical to DBoW2::BowVector (addWeight order + L1 normalize)"
    assert np.array_equal(got["fv_node"], fn) and np.array_equal(got["fv_feature"], ff), "FeatureVector"
    assert abs(bv.sum() - 1.0) < 1e-12


def test_add_if_not_exist(ref):
    word = np.array([5, 3, 5, 9, 3], np.uint32); weight = np.array([1.5, 2.0, 7.0, 0.0, 4.0])
    bw = np.zeros(5, np.uint32); bv = np.zeros(5, np.float64)
    k = ref.ref_bow_add_if_not_exist(_p(word), _p(weight), 5, _p(bw), _p(bv))
    assert bw[:k].tolist() == [3, 5] and bv[:k].tolist() == [2.0, 1.5]
