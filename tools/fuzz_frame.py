"""Randomised parity sweep of the Frame-side steps against the CPU oracle (developer tool): stereo association, grid,
UnprojectStereo and SearchByProjection(Frame, Frame) at random image sizes, scale factors, search radii and predicted poses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g


def run(n_cases, seed0):
    pkg = g.load_package(); orc = g.load_oracle()
    fe, synth = pkg.frontend, pkg.synth
    rng = np.random.default_rng(seed0)
    for k in range(n_cases):
        w = int(rng.integers(400, 1300)); h = int(rng.integers(240, 500)); w = max(w, h + 40)
        scale = float(rng.choice([1.15, 1.2, 1.25, 1.3]))
        lmax = 1 + int(np.floor(np.log(h / 64.0) / np.log(scale)))
        levels = int(rng.integers(3, max(4, min(8, lmax) + 1)))
        nf = int(rng.integers(400, 2000))
        cfg = dict(synth.KITTI_STEREO)
        cfg.update(width=w, height=h, cx=w / 2.0 - 3.5, cy=h / 2.0 + 1.25, fx=float(rng.uniform(400, 800)))
        cfg["fy"] = cfg["fx"]; cfg["bf"] = cfg["fx"] * float(rng.uniform(0.3, 0.6))
        th = float(rng.choice([5.0, 7.0, 10.0, 15.0, 22.0]))
        frames = [synth.stereo_frame(seq=200 + k, t=t, cfg=cfg) for t in range(2)]
        ex = fe.ORBextractor(nf, scale, levels, 20, 7)
        b = fe.Batch(ex, w, h, 4)
        b.extract_host(np.stack([im for (l, r, _) in frames for im in (l, r)]))
        b.stereo_match(2, cfg["bf"], cfg["fx"])
        cam = fe.make_camera(cfg); cam10 = fe.camera_array(cam)
        b.assign_grid(4, cam)
        I = np.eye(4, dtype=np.float32)
        b.unproject(2, 2, cam, np.tile(I, (2, 1, 1)))
        ref = []
        for (l, r, _) in frames:
            oL = orc.Extractor(nf, scale, levels, 20, 7); oR = orc.Extractor(nf, scale, levels, 20, 7)
            kL, dL = oL(l); kR, dR = oR(r)
            ur, dep, _, _ = orc.stereo_matches(oL, oR, kL, dL, kR, dR, cfg["bf"], cfg["fx"])
            ref.append(dict(kp=kL, desc=dL, ur=ur, dep=dep, scale=oL.scale.copy()))
        what = None
        for t in range(2):
            kp, desc, _ = b.download(2 * t)
            ur, dep, _ = b.download_stereo(t)
            n = len(ref[t]["kp"])
            if kp.tobytes() != ref[t]["kp"].tobytes() or not np.array_equal(desc, ref[t]["desc"]): what = "extract frame %d" % t
            elif not (np.array_equal(ur[:n].view(np.uint32), ref[t]["ur"].view(np.uint32)) and np.array_equal(dep[:n].view(np.uint32), ref[t]["dep"].view(np.uint32))):
                what = "stereo frame %d" % t
        if what is None:
            Tcw = I.copy(); Tcw[0, 3] = float(rng.uniform(-0.3, 0.3)); Tcw[2, 3] = float(rng.uniform(-1.0, 1.0))
            xw, fl = b.download_mappoints(0)
            n0 = len(ref[0]["kp"])
            oxw, ovalid = orc.unproject(ref[0]["kp"], ref[0]["dep"], cam10, I)
            if not (np.array_equal(fl[:n0], ovalid) and np.array_equal(xw[:n0].view(np.uint32), oxw.view(np.uint32))): what = "unproject"
            else:
                b.search_by_projection([2], [0], Tcw[None], I[None], cam, th, False, True)
                om, opairs, onm = orc.search_by_projection(ref[1]["kp"], ref[1]["desc"], ref[1]["ur"], ref[0]["kp"], ref[0]["desc"], oxw, ovalid,
                                                           Tcw, I, cam10, ref[1]["scale"], th, False, True)
                m, pairs, nm = b.download_matches(0)
                if nm != onm or not np.array_equal(pairs, opairs) or not np.array_equal(m[:len(ref[1]["kp"])], om): what = "search_by_projection (%d vs %d)" % (nm, onm)
        b.close()
        if what is not None:
            print("MISMATCH case %d (%dx%d scale %.2f levels %d nf %d th %.0f): %s" % (k, w, h, scale, levels, nf, th, what))
            return 1
    print("fuzz_frame: %d cases identical" % n_cases)
    return 0


if __name__ == "__main__":
    sys.exit(run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 5))
