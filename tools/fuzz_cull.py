"""Randomised parity sweep of the dynamic-object cull against the CPU oracle (developer tool): firstSeparate with random box
sets (empty, overlapping, partly outside the image), Separate with H or F (exact or perturbed, so that the static / dynamic
decision goes both ways) and random carried-over box states, UpdateFrame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g


def run(n_cases, seed0):
    import torch
    pkg = g.load_package(); orc = g.load_oracle()
    fe, synth = pkg.frontend, pkg.synth
    cfg = synth.KITTI03_RGBD
    W, H = cfg["width"], cfg["height"]
    rng = np.random.default_rng(seed0)
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    factor = float(np.float32(1.0) / np.float32(cfg["depth_map_factor"]))
    stats = dict(separate_calls=0, ret1=0, matches=0, readmitted=0, boxes=0, empty_boxes_dropped=0, status_changes=0)
    for k in range(n_cases):
        seq = 300 + k
        dt = int(rng.integers(1, 4))
        ts = [0, dt]
        fr = []
        for t in ts:
            rgb, depth, _ = synth.rgbd_frame(seq, t, cfg)
            fr.append((orc.cvt_gray(rgb, 1), depth))
        b = fe.Batch(ex, W, H, 3)
        b.extract_host(np.stack([gr for gr, _ in fr]))
        d_dev = torch.from_numpy(np.stack([d for _, d in fr]).view(np.int16)).cuda()
        b.rgbd_from_u16(d_dev.data_ptr(), W, W * H, 2, factor, cfg["bf"])
        nb = int(rng.integers(0, 9))
        s = 1.01 ** dt
        Hm = np.array([[s, 0, (3.0 * dt - cfg["cx"]) * s + cfg["cx"]], [0, s, -cfg["cy"] * s + cfg["cy"]], [0, 0, 1]], np.float64)
        base = np.zeros((nb, 4))
        base[:, 0] = rng.uniform(-40, W - 60, nb); base[:, 1] = rng.uniform(-30, H - 40, nb)
        base[:, 2] = rng.uniform(2, 260, nb); base[:, 3] = rng.uniform(2, 180, nb)
        if nb > 2 and rng.random() < 0.5: base[1] = base[0] + np.array([15., 8., 0., 0.])       # overlapping pair
        ids = rng.permutation(20)[:nb].astype(np.int32)
        per_frame = []
        for j, t in enumerate(ts):
            bx = base.copy()
            if j == 1:                                   # the boxes follow the scene (some drift away: "dynamic" content changes)
                bx[:, 0] = Hm[0, 0] * bx[:, 0] + Hm[0, 2] + rng.uniform(-6, 6, nb); bx[:, 1] = Hm[1, 1] * bx[:, 1] + Hm[1, 2] + rng.uniform(-4, 4, nb)
                keep = rng.random(nb) < 0.85            # some ids vanish in the current frame
                bx, idj = bx[keep], ids[keep]
            else:
                idj = ids
            per_frame.append((np.maximum(bx, [-1e9, -1e9, 0, 0]), idj.astype(np.int32)))
        ref = []
        for (gr, d), (boxes, idx) in zip(fr, per_frame):
            o = orc.Extractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
            kp, desc = o(gr)
            ur, dep = orc.stereo_from_rgbd(kp, orc.depth_to_f32(d, factor), cfg["bf"])
            r = orc.first_separate(kp, desc, boxes, idx, np.zeros(len(boxes), np.uint8), np.zeros((len(boxes), 2)))
            r["ur"] = ur[r["perm"]]; r["dep"] = dep[r["perm"]]
            ref.append(r)
            stats["boxes"] += len(r["boxes"]); stats["empty_boxes_dropped"] += len(boxes) - len(r["boxes"])
        b.first_separate([0, 1], [p[0] for p in per_frame], [p[1] for p in per_frame])
        what = None
        for slot, r in enumerate(ref):
            gq = b.download_boxes(slot)
            kp, desc, _ = b.download(slot)
            Ns = r["Ns"]
            ok = (gq["n_static"] == Ns and gq["n_all"] == Ns + r["Nd"] and gq["nb"] == len(r["boxes"]) and np.array_equal(gq["boxes"], r["boxes"])
                  and np.array_equal(gq["box_idx"], r["box_idx"]) and np.array_equal(gq["boxStart"], r["boxStart"])
                  and np.array_equal(gq["boxItems"] + Ns, r["boxItems"]) and kp.tobytes() == r["kp"][:Ns].tobytes() and np.array_equal(desc, r["desc"][:Ns]))
            dk, dd, dur, ddep = b.download_dynamic(slot)
            ok = ok and dk.tobytes() == r["kp"][Ns:].tobytes() and np.array_equal(dd, r["desc"][Ns:])
            if not ok: what = "firstSeparate slot %d" % slot
        if what is None and len(ref[1]["box_idx"]) > 0:
            cur, rf = ref[1], ref[0]
            flag = int(rng.integers(1, 3))
            Hp = Hm.copy()
            if rng.random() < 0.4: Hp[0, 2] += rng.uniform(-6, 6); Hp[1, 2] += rng.uniform(-6, 6)      # a wrong model: matches become "dynamic"
            if flag == 1: M = Hp.astype(np.float32)
            else:
                e = np.array([cfg["cx"], cfg["cy"], 1.0])
                exm = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
                M = (exm @ Hp).astype(np.float32)
            nl = int(rng.integers(0, 5))
            last_idx = rng.permutation(20)[:nl].astype(np.int32)
            last_status = rng.integers(-1, 3, nl).astype(np.int32)
            cur_boxes = b.download_boxes(1)
            oret, osc, ods, odyn, omt = orc.separate(M, flag, dict(kp=cur["kp"], desc=cur["desc"], boxStart=cur["boxStart"], boxItems=cur["boxItems"], box_idx=cur["box_idx"]),
                                                     dict(kp=rf["kp"], desc=rf["desc"], boxStart=rf["boxStart"], boxItems=rf["boxItems"], box_idx=rf["box_idx"]),
                                                     last_idx, last_status, cur_boxes["box_status"])
            b.separate([1], [0], M[None], [flag], [last_idx], [last_status])
            ret, ds, dyn, mt = b.download_separate(0)
            stats["separate_calls"] += 1; stats["ret1"] += int(oret == 1); stats["matches"] += len(omt)
            stats["status_changes"] += int((osc != cur_boxes["box_status"]).sum())
            nbc = len(cur["box_idx"])
            if not (ret == oret and np.array_equal(ds[:nbc + 1], ods) and np.array_equal(mt, omt) and np.array_equal(dyn, odyn)
                    and np.array_equal(b.download_boxes(1)["box_status"], osc)):
                what = "Separate (flag %d, ret %d vs %d)" % (flag, ret, oret)
            else:
                n_before = int(b.counts(2)[1])
                app = orc.update_frame(cur["kp"], cur["boxStart"], cur["boxItems"], ods, odyn)
                stats["readmitted"] += len(app)
                b.update_frame(only_if_static=False)
                n_after = int(b.counts(2)[1])
                kp, desc, _ = b.download(1)
                exp_kp = np.concatenate([cur["kp"][:cur["Ns"]], cur["kp"][app]]); exp_desc = np.concatenate([cur["desc"][:cur["Ns"]], cur["desc"][app]])
                if not (n_after == n_before + len(app) and kp.tobytes() == exp_kp.tobytes() and np.array_equal(desc, exp_desc)):
                    what = "UpdateFrame (%d + %d vs %d)" % (n_before, len(app), n_after)
        b.close()
        if what is not None:
            print("MISMATCH case %d (seq %d, dt %d, %d boxes): %s" % (k, seq, dt, nb, what))
            return 1
    print("fuzz_cull: %d cases identical" % n_cases, stats)
    return 0


if __name__ == "__main__":
    sys.exit(run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 3))
