"""GPU parity of the bag-of-words path: vocabulary loading / packing, Frame::ComputeBoW (DBoW2 transform, BowVector,
FeatureVector) and ORBmatcher::SearchByBoW, bit-exact (f64 values included) against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seq(gpu, fe, synth):
    cfg = synth.KITTI_STEREO
    T = 3
    frames = [synth.stereo_frame(seq=4, t=t) for t in range(T)]
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    b = fe.Batch(ex, cfg["width"], cfg["height"], T)
    b.extract_host(np.stack([l for (l, r, _) in frames]))
    ref = []
    for t in range(T):
        kp, desc, _ = b.download(t)
        ref.append(dict(kp=kp, desc=desc))
    yield dict(b=b, ref=ref, T=T)
    b.close()


def _tuned_vocabulary(synth, ref, L, seed):
    """A synthetic tree whose level-1 centres are real descriptors of the sequence, so that features spread over it."""
    voc = synth.vocabulary(k=10, L=L, seed=seed)
    rng = np.random.default_rng(seed)
    d = ref[0]["desc"]
    first = np.nonzero(voc["parent"] == 0)[0]
    voc["desc"][first] = d[rng.choice(len(d), len(first), replace=False)]
    # re-derive deeper levels from their (new) parents so that the descent stays meaningful
    for i in range(len(voc["parent"])):
        p = voc["parent"][i]
        if p > 0:
            base = voc["desc"][p - 1].copy()
            for bit in rng.integers(0, 256, 24 >> min(3, i % 4)):
                base[bit >> 3] ^= np.uint8(1 << (bit & 7))
            voc["desc"][i] = base
    return voc


def test_vocabulary_load_pack_adopt(fe, orc, synth, tmp_path):
    import torch
    voc = synth.vocabulary(k=10, L=3, seed=21)
    p = tmp_path / "voc.txt"
    synth.write_vocabulary_text(voc, p)
    A = fe.Vocabulary.load_text(p); B = fe.Vocabulary.from_nodes(voc); O = orc.Vocabulary.load_text(p)
    assert A.info() == B.info() == O.info()
    na, nb, no = A.nodes(), B.nodes(), O.nodes()
    for k in no:
        assert np.array_equal(na[k], no[k]) and np.array_equal(nb[k], no[k]), k
    ptr, nbytes = A.packed_device()
    assert nbytes == fe.Vocabulary.packed_bytes(A.info()["n_nodes"])
    # what a non-zero rank does after the broadcast: adopt a buffer it allocated itself
    mine = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    mine.copy_(fe.as_torch_u8(ptr, nbytes))
    torch.cuda.synchronize()
    D = fe.Vocabulary.from_packed_device(mine.data_ptr(), nbytes)
    assert D.info() == A.info()
    nd = D.nodes()
    for k in no:
        assert np.array_equal(nd[k], no[k]), k
    with pytest.raises(fe.SdError):
        fe.Vocabulary.from_packed_device(mine.data_ptr() + 256, nbytes - 256)       # not a packed vocabulary
    bad = tmp_path / "bad.txt"
    bad.write_text("30 6 0 0\n")
    with pytest.raises(fe.SdError):
        fe.Vocabulary.load_text(bad)
    with pytest.raises(fe.SdError):
        fe.Vocabulary.load_text(tmp_path / "missing.txt")


@pytest.mark.parametrize("L,levelsup,scoring,weighting", [(4, 2, 0, 0), (3, 4, 0, 0), (3, 1, 1, 1), (3, 1, 5, 0), (3, 2, 0, 2)])
def test_compute_bow(seq, fe, orc, synth, L, levelsup, scoring, weighting):
    b, ref, T = seq["b"], seq["ref"], seq["T"]
    voc = _tuned_vocabulary(synth, ref, L, seed=10 * L + levelsup)
    voc["scoring"], voc["weighting"] = scoring, weighting
    V = fe.Vocabulary.from_nodes(voc); O = orc.Vocabulary.from_nodes(voc)
    b.compute_bow(V, list(range(T)), levelsup)
    for t in range(T):
        g = b.download_bow(t)
        word, w, nid = O.transform(ref[t]["desc"], levelsup)
        assert np.array_equal(g["f_word"], word) and np.array_equal(g["f_node"], nid)
        assert np.array_equal(g["f_weight"].view(np.uint64), w.view(np.uint64))
        o = O.compute_bow(ref[t]["desc"], levelsup)
        assert len(o["word"]) > 50
        assert np.array_equal(g["word"], o["word"])
        assert np.array_equal(g["value"].view(np.uint64), o["value"].view(np.uint64)), "BowVector values must match bit for bit"
        assert np.array_equal(g["fv_node"], o["fv_node"]) and np.array_equal(g["fv_feature"], o["fv_feature"])
    V.close()


@pytest.mark.parametrize("L,levelsup,nnratio,valid_frac,ori", [(4, 2, 0.7, 1.0, True), (4, 3, 0.75, 0.7, True), (3, 4, 0.9, 0.9, False)])
def test_search_by_bow(seq, fe, orc, synth, L, levelsup, nnratio, valid_frac, ori):
    """Frame t against keyframe t-1 (TrackReferenceKeyFrame uses ORBmatcher(0.7, true), Relocalization 0.75).  levelsup = 4 on an
    L = 3 tree puts every feature under the root: one node of ~2000 x ~2000 candidates."""
    import torch
    b, ref, T = seq["b"], seq["ref"], seq["T"]
    voc = _tuned_vocabulary(synth, ref, L, seed=7 * L + levelsup)
    V = fe.Vocabulary.from_nodes(voc); O = orc.Vocabulary.from_nodes(voc)
    b.compute_bow(V, list(range(T)), levelsup)
    rng = np.random.default_rng(int(100 * nnratio) + levelsup)
    npairs = T - 1
    valid = (rng.random((npairs, b.cap)) < valid_frac).astype(np.uint8)
    d_valid = torch.from_numpy(valid).cuda()
    b.search_by_bow(list(range(npairs)), [p + 1 for p in range(npairs)], nnratio, ori, d_kf_valid=d_valid.data_ptr())
    total = 0
    for p in range(npairs):
        kf, fr = ref[p], ref[p + 1]
        bk, bf = O.compute_bow(kf["desc"], levelsup), O.compute_bow(fr["desc"], levelsup)
        om, onm = orc.search_by_bow(kf["kp"], kf["desc"], valid[p, :len(kf["kp"])], bk, fr["kp"], fr["desc"], bf, nnratio, ori)
        m, pairs, nm = b.download_matches(p)
        assert nm == onm, "nmatches pair %d: %d vs %d" % (p, nm, onm)
        assert np.array_equal(m[:len(fr["kp"])], om), "vpMapPointMatches, pair %d" % p
        total += onm
    assert total > 30 * npairs, "expected plenty of matches, got %d" % total
    V.close()


def test_randomised_vocabularies(gpu):
    """8 fixed draws of tools/fuzz_bow.py (branching factor 2..20, depth, early leaves, zero-weight words, scoring / weighting,
    levelsup 0..L+1, SearchByBoW ratio / orientation / valid mask): identical to the oracle, f64 values bit for bit.
    (40 draws were run when written.)"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_bow", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_bow.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert m.run(8, 5) == 0
