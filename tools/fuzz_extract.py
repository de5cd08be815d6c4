"""Randomised parity sweep of the extractor against the CPU oracle (developer tool; a few fixed draws are in tests/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

def draw(rng):
    w = int(rng.integers(160, 1400)); h = int(rng.integers(120, 520))
    scale = float(rng.choice([1.1, 1.15, 1.2, 1.25, 1.3, 1.4, 1.5, 1.7, 2.0]))
    w = max(w, h + 1)                                                          # landscape (portrait levels are undefined in the reference)
    lmax = 1 + int(np.floor(np.log(min(w, h) / 64.0) / np.log(scale)))         # smallest level keeps at least one 30-px cell row
    levels = int(rng.integers(1, max(2, min(8, lmax) + 1)))
    nf = int(rng.integers(100, 2500)) if levels > 2 else int(rng.integers(100, 1200))
    ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini + 1))
    kind = str(rng.choice(["texture", "mixed", "noise"], p=[0.6, 0.3, 0.1]))
    return w, h, scale, levels, nf, ini, mn, kind

def run(n_cases, seed0):
    pkg = g.load_package(); orc = g.load_oracle()
    fe, synth = pkg.frontend, pkg.synth
    rng = np.random.default_rng(seed0)
    ok = skipped = 0
    for k in range(n_cases):
        w, h, scale, levels, nf, ini, mn, kind = draw(rng)
        img = synth.random_image(w, h, 1000 + k, kind)
        try:
            ex = fe.ORBextractor(nf, scale, levels, ini, mn)
            b = fe.Batch(ex, w, h, 1)
        except fe.SdError as e:
            skipped += 1
            print("case %d skipped (%s): %s" % (k, (w, h, scale, levels, nf), str(e)[:90]))
            continue
        b.extract_host(img[None])
        o = orc.Extractor(nf, scale, levels, ini, mn)
        rk, rd = o(img)
        kp, desc, per_level = b.download(0)
        same = np.array_equal(per_level, o.per_level) and kp.tobytes() == rk.tobytes() and np.array_equal(desc, rd)
        if not same:
            lv = [l for l in range(levels) if not np.array_equal(b.pyramid(0, l), o.pyramid(l))]
            print("MISMATCH case %d: %r  per-level %r vs %r; pyramid levels differing: %r" % (k, (w, h, scale, levels, nf, ini, mn, kind), per_level, o.per_level, lv))
            b.close()
            return 1
        ok += 1
        b.close()
    print("fuzz: %d cases identical, %d skipped as unsupported geometry" % (ok, skipped))
    return 0

if __name__ == "__main__":
    sys.exit(run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 7))
