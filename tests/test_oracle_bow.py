"""CPU tests of the bag-of-words oracle (oracle/bow_oracle.inc): the DBoW2 text format, tree descent, BowVector /
FeatureVector construction and ORBmatcher::SearchByBoW on hand-made cases with known answers."""
import os
import numpy as np
import pytest


def _bits(*ones):
    d = np.zeros(32, np.uint8)
    for b in ones:
        d[b >> 3] |= 1 << (b & 7)
    return d


@pytest.fixture(scope="module")
def tiny(orc):
    """k = 2, L = 2.  Node ids: 1,2 under the root; 3,4 under 1; 5 (leaf, early) is node 2 itself.
    line: parent isLeaf desc weight"""
    lines = dict(k=2, L=2, scoring=0, weighting=0,
                 parent=np.array([0, 0, 1, 1], np.int32),
                 is_leaf=np.array([0, 1, 1, 1], np.uint8),
                 desc=np.stack([_bits(), _bits(*range(64, 192)), _bits(0, 1), _bits(2, 3, 4, 5)]),
                 weight=np.array([0.0, 2.5, 1.25, 0.0], np.float64))
    return lines, orc.Vocabulary.from_nodes(lines)


def test_tree_shape_and_word_ids(tiny):
    lines, V = tiny
    assert V.info() == dict(k=2, L=2, scoring=0, weighting=0, n_nodes=5, n_words=3)
    n = V.nodes()
    assert n["parent"].tolist() == [0, 0, 0, 1, 1] and n["n_children"].tolist() == [2, 2, 0, 0, 0]
    assert n["word_id"].tolist() == [-1, -1, 0, 1, 2]            # words numbered in file order (m_words.size() at the line)


def test_transform_descent_ties_and_early_leaf(tiny):
    _, V = tiny
    f = np.stack([_bits(),                  # 0 from node 1, 128 from node 2 -> node 1; children: 2 vs 4 bits -> node 3 (word 1)
                  _bits(*range(64, 192)),   # exactly node 2 -> leaf at level 1 (word 0)
                  _bits(*range(64, 128)),   # 64 from node 1 AND 64 from node 2: tie -> first child (node 1); then 66 vs 68 -> node 3
                  _bits(2, 3, 4, 5)])       # node 1 (4 vs 132), then node 4 (distance 0), weight 0 = stop word
    word, w, nid = V.transform(f, levelsup=0)             # nid at level L - 0 = 2
    assert word.tolist() == [1, 0, 1, 2] and w.tolist() == [1.25, 2.5, 1.25, 0.0]
    assert nid.tolist() == [3, 2, 3, 4]                    # feature 1 never reaches level 2: the leaf it reached (spec Q12)
    word, w, nid = V.transform(f, levelsup=1)             # nid at level 1
    assert nid.tolist() == [1, 2, 1, 1]
    word, w, nid = V.transform(f, levelsup=4)             # L - levelsup <= 0 -> root
    assert nid.tolist() == [0, 0, 0, 0]


def test_bow_and_feature_vectors(tiny):
    _, V = tiny
    f = np.stack([_bits(), _bits(*range(64, 192)), _bits(*range(64, 128)), _bits(2, 3, 4, 5), _bits(0)])
    b = V.compute_bow(f, levelsup=1)
    # words: f0 -> 1, f1 -> 0, f2 -> 1, f3 -> stop word (skipped), f4 -> 1 (1 bit from node 3)
    assert b["word"].tolist() == [0, 1]
    s = 2.5 + ((1.25 + 1.25) + 1.25)
    assert b["value"].tolist() == [2.5 / s, ((1.25 + 1.25) + 1.25) / s]
    assert b["fv_node"].tolist() == [1, 1, 1, 2] and b["fv_feature"].tolist() == [0, 2, 4, 1]


def test_text_round_trip_and_header_check(orc, synth, tmp_path):
    voc = synth.vocabulary(k=10, L=3, seed=11)
    p = tmp_path / "voc.txt"
    synth.write_vocabulary_text(voc, p)
    A, B = orc.Vocabulary.from_nodes(voc), orc.Vocabulary.load_text(p)
    na, nb = A.nodes(), B.nodes()
    for k in na:
        assert np.array_equal(na[k], nb[k]), k
    assert A.info() == B.info() and A.info()["n_nodes"] == len(voc["parent"]) + 1
    bad = tmp_path / "bad.txt"
    bad.write_text("30 6 0 0\n0 1 " + " ".join(["0"] * 32) + " 1.0\n")       # k > 20: "not a correct text file" (:1359-1363)
    with pytest.raises(ValueError):
        orc.Vocabulary.load_text(bad)
    with pytest.raises(ValueError):
        orc.Vocabulary.load_text(tmp_path / "missing.txt")


def test_l1_score_of_normalised_vectors(orc, synth):
    voc = synth.vocabulary(k=10, L=3, seed=5)
    V = orc.Vocabulary.from_nodes(voc)
    rng = np.random.default_rng(1)
    d1 = rng.integers(0, 256, (300, 32), dtype=np.uint8); d2 = d1.copy(); d2[:150] = rng.integers(0, 256, (150, 32), dtype=np.uint8)
    a, b = V.compute_bow(d1), V.compute_bow(d2)
    assert abs(a["value"].sum() - 1) < 1e-12 and np.all(np.diff(a["word"].astype(np.int64)) > 0)
    assert abs(orc.bow_score_l1(a, a) - 1.0) < 1e-12                 # L1Scoring::score(v, v) = 1 for L1-normalised vectors
    s = orc.bow_score_l1(a, b)
    assert 0.2 < s < 0.9 and abs(s - orc.bow_score_l1(b, a)) < 1e-15


def test_search_by_bow_known_answer(orc):
    KP = orc.KP_DTYPE
    kK = np.zeros(3, KP); kF = np.zeros(4, KP)
    kK["angle"] = [10, 20, 200]; kF["angle"] = [12, 22, 100, 14]
    dK = np.stack([_bits(), _bits(*range(0, 8)), _bits(*range(100, 130))])
    dF = np.stack([_bits(0), _bits(*range(0, 9)), _bits(*range(100, 131)), _bits(0, 1, 2)])
    # one shared node (7) holds everything; keyframe features are walked in order 0, 1, 2
    bK = dict(fv_node=np.array([7, 7, 7], np.uint32), fv_feature=np.array([0, 1, 2], np.uint32))
    bF = dict(fv_node=np.array([7, 7, 7, 7], np.uint32), fv_feature=np.array([0, 1, 2, 3], np.uint32))
    m, nm = orc.search_by_bow(kK, dK, np.ones(3, np.uint8), bK, kF, dF, bF, 0.7, checkOrientation=False)
    # kf0: distances 1, 9, 31, 3 -> best 1 (f0), second 3: 1 < 2.1 ok.  kf1: f0 taken; 1 (f1), 29, 5 (f3): 1 < 3.5 ok.
    # kf2: f0, f1 taken; 1 (f2), 33 (f3) ok.
    assert m.tolist() == [0, 1, 2, -1] and nm == 3
    m, nm = orc.search_by_bow(kK, dK, np.array([1, 0, 1], np.uint8), bK, kF, dF, bF, 0.7, checkOrientation=False)
    assert m.tolist() == [0, -1, 2, -1] and nm == 2                  # a keyframe feature without a map point is skipped
    # rotation histogram: rot = 358 (bin 12), 358 (bin 12), 100 (bin 3): max1 = 2, max2 = 1 >= 0.1*2 -> both bins kept
    m, nm = orc.search_by_bow(kK, dK, np.ones(3, np.uint8), bK, kF, dF, bF, 0.7, checkOrientation=True)
    assert m.tolist() == [0, 1, 2, -1] and nm == 3
    # different nodes never match
    bF2 = dict(fv_node=np.array([8, 8, 8, 8], np.uint32), fv_feature=np.array([0, 1, 2, 3], np.uint32))
    m, nm = orc.search_by_bow(kK, dK, np.ones(3, np.uint8), bK, kF, dF, bF2, 0.7)
    assert nm == 0 and (m == -1).all()
