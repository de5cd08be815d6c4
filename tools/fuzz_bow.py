"""Randomised parity sweep of the bag-of-words path against the CPU oracle (developer tool): vocabularies with random branching
factor, depth, early leaves, scoring and weighting; Frame::ComputeBoW at random levelsup; SearchByBoW at random ratios."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g


def run(n_cases, seed0):
    import torch
    pkg = g.load_package(); orc = g.load_oracle()
    fe, synth = pkg.frontend, pkg.synth
    cfg = synth.KITTI_STEREO
    rng = np.random.default_rng(seed0)
    T = 2
    frames = [synth.stereo_frame(seq=400 + seed0, t=t) for t in range(T)]
    nf = int(rng.integers(600, 2000))
    ex = fe.ORBextractor(nf, cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    b = fe.Batch(ex, cfg["width"], cfg["height"], T)
    b.extract_host(np.stack([l for (l, r, _) in frames]))
    ref = []
    for t in range(T):
        kp, desc, _ = b.download(t)
        ref.append(dict(kp=kp, desc=desc))
    stats = dict(words=0, matches=0)
    for k in range(n_cases):
        kk = int(rng.choice([2, 3, 5, 8, 10, 12, 16, 20]))
        L = int(rng.integers(2, 6)) if kk <= 5 else int(rng.integers(2, 4))
        voc = synth.vocabulary(k=kk, L=L, seed=1000 + k, early_leaf_frac=float(rng.choice([0.0, 0.02, 0.2])), stop_frac=float(rng.choice([0.0, 0.01, 0.1])))
        d = ref[0]["desc"]
        first = np.nonzero(voc["parent"] == 0)[0]
        voc["desc"][first] = d[rng.choice(len(d), len(first), replace=False)]
        voc["scoring"], voc["weighting"] = int(rng.choice([0, 1, 2, 3, 4, 5])), int(rng.integers(0, 4))
        levelsup = int(rng.integers(0, L + 2))
        V = fe.Vocabulary.from_nodes(voc); O = orc.Vocabulary.from_nodes(voc)
        b.compute_bow(V, list(range(T)), levelsup)
        what = None
        bows = []
        for t in range(T):
            gq = b.download_bow(t)
            word, w, nid = O.transform(ref[t]["desc"], levelsup)
            o = O.compute_bow(ref[t]["desc"], levelsup)
            bows.append(o)
            stats["words"] += len(o["word"])
            if not (np.array_equal(gq["f_word"], word) and np.array_equal(gq["f_node"], nid) and np.array_equal(gq["f_weight"].view(np.uint64), w.view(np.uint64))):
                what = "transform frame %d" % t
            elif not (np.array_equal(gq["word"], o["word"]) and np.array_equal(gq["value"].view(np.uint64), o["value"].view(np.uint64))):
                what = "BowVector frame %d" % t
            elif not (np.array_equal(gq["fv_node"], o["fv_node"]) and np.array_equal(gq["fv_feature"], o["fv_feature"])):
                what = "FeatureVector frame %d" % t
        if what is None:
            nnratio = float(rng.choice([0.6, 0.7, 0.75, 0.9])); ori = bool(rng.integers(0, 2))
            valid = (rng.random((1, b.cap)) < float(rng.choice([1.0, 0.8, 0.3]))).astype(np.uint8)
            d_valid = torch.from_numpy(valid).cuda()
            b.search_by_bow([0], [1], nnratio, ori, d_kf_valid=d_valid.data_ptr())
            om, onm = orc.search_by_bow(ref[0]["kp"], ref[0]["desc"], valid[0, :len(ref[0]["kp"])], bows[0], ref[1]["kp"], ref[1]["desc"], bows[1], nnratio, ori)
            m, pairs, nm = b.download_matches(0)
            stats["matches"] += onm
            if nm != onm or not np.array_equal(m[:len(ref[1]["kp"])], om): what = "SearchByBoW (%d vs %d)" % (nm, onm)
        V.close()
        if what is not None:
            print("MISMATCH case %d (k %d, L %d, levelsup %d, scoring %d, weighting %d): %s" % (k, kk, L, levelsup, voc["scoring"], voc["weighting"], what))
            b.close()
            return 1
    b.close()
    print("fuzz_bow: %d cases identical" % n_cases, stats)
    return 0


if __name__ == "__main__":
    sys.exit(run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 3))
