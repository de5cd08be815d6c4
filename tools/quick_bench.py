"""Developer timing loop (not the driver's bench.py): per-kernel hipEvent times for a stereo batch."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); fe, synth = pkg.frontend, pkg.synth
import torch
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = synth.KITTI_STEREO
frames = [synth.stereo_frame(seq=0, t=t % 8) for t in range(min(nf, 8))]
imgs = np.stack([im for k in range(nf) for im in frames[k % len(frames)][:2]])
d = torch.from_numpy(imgs).cuda()
ex = fe.ORBextractor(cfg["n_features"], 1.2, 8, cfg["ini_th_fast"], cfg["min_th_fast"])
b = fe.Batch(ex, cfg["width"], cfg["height"], 2 * nf)
W, H = cfg["width"], cfg["height"]
def step():
    b.extract_device(d.data_ptr(), W, W * H, 2 * nf)
    b.stereo_match(nf, cfg["bf"], cfg["fx"])
for _ in range(3): step()
b.sync()
t = time.time(); K = 10
for _ in range(K): step()
b.sync(); dt = (time.time() - t) / K
print("frames/batch %d  ms/batch %.3f  stereo fps %.0f" % (nf, dt * 1e3, nf / dt))
b.set_profiling(True); b.reset_kernel_times()
for _ in range(5): step()
b.sync()
tot = 0
for k, (ms, n) in b.kernel_times().items():
    if n: print("  %-18s %8.3f ms/batch  (%d launches)" % (k, ms / 5, n)); tot += ms / 5
print("  sum %.3f ms" % tot)
print("counts", b.counts(4))
