"""Developer timing of the detector (images/s, achieved TFLOP/s vs the f16 MFMA dense peak)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); synth = pkg.synth
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PREC = sys.argv[2] if len(sys.argv) > 2 else "f16"
PEAK = {"f16": 2500.0, "f32": 157.3, "f32w": 157.3, "f32x3": 157.3}[PREC]
layers, anchors = pkg.yolo.v3_layers()
payload, _ = pkg.yolo.synth_weights(layers, seed=3)
d = pkg.yolo.Detector(layers, anchors, 640, 480, max_batch=B, precision=PREC)
d.load_weights(payload)
cfg = synth.KITTI03_RGBD
imgs = np.stack([np.ascontiguousarray(synth.rgbd_frame(6, t % 4, cfg)[0][:, :, ::-1]) for t in range(B)])
dev = torch.from_numpy(imgs).cuda()
H, W = imgs.shape[1:3]
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    d.forward_device(dev.data_ptr(), W, H, W * 3, W * H * 3, B, 0.5, st)
torch.cuda.synchronize()
K = 20 if PREC == "f16" else 5
t = time.time()
for _ in range(K):
    d.forward_device(dev.data_ptr(), W, H, W * 3, W * H * 3, B, 0.5, st)
torch.cuda.synchronize()
dt = (time.time() - t) / K
fl = d.flops()
if PREC == "f32x3":
    print("executed: %.2f G bf16 MFMA FLOPs + %.2f G f32 MFMA FLOPs per image: matrix-pipe time %.1f %% of the batch time"
          % (d.mfma_flops_bf16() / 1e9, d.mfma_flops() / 1e9, (d.mfma_flops_bf16() / 2500e12 + d.mfma_flops() / 157.3e12) * B / dt * 100))
if PREC == "f32w":
    print("executed MFMA FLOPs %.2f G of %.2f G nominal per image: %.1f TFLOP/s executed" % (d.mfma_flops() / 1e9, fl / 1e9, d.mfma_flops() * B / dt / 1e12))
print("%s batch %d: %.2f ms/batch, %.1f images/s, %.1f TFLOP/s (%.1f%% of the %.1f TFLOP/s dense %s MFMA peak)" % (PREC, B, dt * 1e3, B / dt, fl * B / dt / 1e12, fl * B / dt / (PEAK * 1e12) * 100, PEAK, PREC))
