"""Detector modes against each other on many images (GPU): post-NMS box sets of SD_YOLO_F32W (Winograd) vs SD_YOLO_F32 (direct sums).
Both are f32 evaluations of the same network that differ in summation only, so the box sets (int-truncated corners, class ids, order)
should agree except where a coordinate or a score sits within ~1e-6 of a decision boundary; this counts how often that happens.
(The parity tests compare each mode with a torch-fp32 forward on the CPU: tests/test_gpu_yolo.py; this sweep is the wider, cheaper net.)
With a third argument N, the first N images also go through the torch-fp32 forward of the oracle (CPU, ~0.2 s per image) and BOTH modes are
counted against it -- the flip rate of the direct mode is the yardstick for the Winograd mode's.
usage: python tools/fuzz_yolo.py [n_images=512] [seed=0] [n_torch=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); synth = pkg.synth
import torch

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 512
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_torch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B = 32
layers, anchors = pkg.yolo.v3_layers()
payload, per = pkg.yolo.synth_weights(layers, seed=3)
if n_torch:
    yo = g.load_yolo_oracle(); orc = g.load_oracle()
    torch.set_num_threads(min(16, torch.get_num_threads() or 8))
t_same = [0, 0]; t_boxes = 0; t_found = [0, 0]
da = pkg.yolo.Detector(layers, anchors, 640, 480, max_batch=B, precision="f32")
OTHER = os.environ.get("FUZZ_YOLO_MODE", "f32w")          # the mode compared with the direct f32 one: f32w (Winograd) or f32x3 (bf16 limbs)
db = pkg.yolo.Detector(layers, anchors, 640, 480, max_batch=B, precision=OTHER)
da.load_weights(payload); db.load_weights(payload)
cfg = synth.KITTI_STEREO
W, H = cfg["width"], cfg["height"]
rng = np.random.default_rng(seed)
same = boxes = boxes_same = 0
worst_conf = 0.0
diff = []
for b0 in range(0, n_img, B):
    imgs = []
    for k in range(B):
        left, _, _ = synth.stereo_frame_dyn(int(rng.integers(0, 200)), int(rng.integers(0, 12)), cfg)
        gq = left.astype(np.int16)
        img = np.stack([np.clip(gq + rng.integers(-6, 7, gq.shape), 0, 255) for _ in range(3)], -1).astype(np.uint8)
        if rng.integers(0, 2):
            img = np.ascontiguousarray(img[:, ::-1])
        imgs.append(img)
    dev = torch.from_numpy(np.stack(imgs)).cuda()
    da.forward_device(dev.data_ptr(), W, H, W * 3, W * H * 3, B, 0.5)
    db.forward_device(dev.data_ptr(), W, H, W * 3, W * H * 3, B, 0.5)
    ra = da.boxes_batch(B, W, H); rb = db.boxes_batch(B, W, H)
    for k in range(B):
        ok = np.array_equal(ra[k][0], rb[k][0]) and np.array_equal(ra[k][1], rb[k][1])
        same += ok
        boxes += len(ra[k][0])
        boxes_same += sum(1 for r in ra[k][0] if any(np.array_equal(r, q) for q in rb[k][0]))
        if ok and len(ra[k][2]):
            worst_conf = max(worst_conf, float(np.max(np.abs(ra[k][2] - rb[k][2]))))
        if not ok:
            diff.append((b0 + k, len(ra[k][0]), len(rb[k][0])))
        if b0 + k < n_torch:
            ref = yo.torch_forward(layers, per, yo.blob_from_image(imgs[k], 640, 480, orc.resize_linear))
            rows = np.concatenate([yo.region_decode(ref[li - 1][0].numpy().transpose(1, 2, 0), list(layers[li]["mask"]), anchors, 640, 480) for li in (82, 94, 106)])
            eb, ec, ef = yo.postprocess(rows, W, H, 0.5, 0.4)
            t_boxes += len(eb)
            for m, r in enumerate((ra, rb)):
                t_same[m] += np.array_equal(r[k][0], eb) and np.array_equal(r[k][1], ec)
                t_found[m] += sum(1 for q in eb if any(np.array_equal(q, z) for z in r[k][0]))
da.close(); db.close()
print("fuzz_yolo: %d images, %s against f32: box sets identical on %d; %d of %d f32 boxes found bit for bit in the %s set; largest confidence difference on identical sets %.3g"
      % (n_img, OTHER, same, boxes_same, boxes, OTHER, worst_conf))
if n_torch:
    print("against the torch-fp32 forward on the first %d images: f32 box sets equal on %d (%d of %d boxes), %s on %d (%d of %d boxes)"
          % (n_torch, t_same[0], t_found[0], t_boxes, OTHER, t_same[1], t_found[1], t_boxes))
if diff:
    print("differing images (index, boxes f32, boxes %s):" % OTHER, diff[:20])
