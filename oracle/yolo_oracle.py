"""TEST INFRASTRUCTURE — reference implementation of the detector path for the parity tests.

The convolution stack is checked against a plain PyTorch fp32 CPU forward (the task's rule for floating-point
kernels); blobFromImage, the cv::dnn region layer and yolov3Segment::postprocess_ (src/yolo.cc:151-206) are restated
in numpy [OpenCV-recall].  yolov3.weights does not exist offline: PARITY UNPINNED, synthetic weights only."""
import numpy as np

CONV, SHORTCUT, ROUTE, UPSAMPLE, YOLO = 0, 1, 2, 3, 4


def blob_from_image(bgr, net_w, net_h, resize_linear):
    """blobFromImage(img, 1/255, Size(net_w, net_h), 0, swapRB=True, crop=False) -> (3, H, W) f32 in RGB order."""
    ch = [resize_linear(np.ascontiguousarray(bgr[:, :, c]), net_w, net_h) for c in range(3)]
    rgb = np.stack([ch[2], ch[1], ch[0]]).astype(np.float32)
    return rgb * np.float32(1 / 255.0)


def torch_forward(layers, per_conv, blob, dtype=None):
    """fp32 forward of the Darknet graph; returns the list of layer outputs (torch tensors, NCHW).  dtype=torch.float64: the same graph on
    the same f32 weights / blob widened to double -- the ground truth the f32-class detector modes are ranked against (every mode, and torch's own
    fp32 forward, is a differently ordered rounding of THIS)."""
    import torch
    import torch.nn.functional as F
    dt = dtype or torch.float32
    x = torch.from_numpy(blob)[None].to(dt)
    outs = []
    with torch.no_grad():
        for i, l in enumerate(layers):
            t = int(l["type"])
            if t == CONV:
                p = per_conv[i]
                w = torch.from_numpy(p["w"]).to(dt)
                x = F.conv2d(x, w, None, stride=int(l["stride"]), padding=int(l["size"]) // 2)
                if l["batch_normalize"]:
                    if dt == torch.float32:
                        s = torch.from_numpy(p["gamma"] / np.sqrt(p["var"] + np.float32(1e-6)))
                    else:
                        s = torch.from_numpy(p["gamma"]).to(dt) / torch.sqrt(torch.from_numpy(p["var"]).to(dt) + 1e-6)
                    x = (x - torch.from_numpy(p["mean"]).to(dt)[None, :, None, None]) * s[None, :, None, None] + torch.from_numpy(p["beta"]).to(dt)[None, :, None, None]
                else:
                    x = x + torch.from_numpy(p["bias"]).to(dt)[None, :, None, None]
                if l["leaky"]:
                    x = F.leaky_relu(x, 0.1)
            elif t == SHORTCUT:
                f = int(l["from"][0]); x = outs[i + f if f < 0 else f] + x
            elif t == ROUTE:
                srcs = [outs[i + int(f) if f < 0 else int(f)] for f in l["from"][:int(l["nfrom"])]]
                x = torch.cat(srcs, 1) if len(srcs) > 1 else srcs[0]
            elif t == UPSAMPLE:
                x = F.interpolate(x, scale_factor=2, mode="nearest")
            elif t == YOLO:
                pass
            outs.append(x)
    return outs


def region_decode(head_hwc, mask, anchors, net_w, net_h, thresh=0.001):
    """cv::dnn RegionLayer (YOLOv3: logistic, classes scaled by objectness) for one head: (H, W, 255) -> rows (H*W*3, 85)."""
    H, W, _ = head_hwc.shape
    t = head_hwc.astype(np.float32).reshape(H, W, 3, 85)
    sig = lambda v: (np.float32(1) / (np.float32(1) + np.exp(-v, dtype=np.float32))).astype(np.float32)
    xs = np.arange(W, dtype=np.float32)[None, :, None]; ys = np.arange(H, dtype=np.float32)[:, None, None]
    aw = np.array([anchors[2 * m] for m in mask], np.float32)[None, None, :]
    ah = np.array([anchors[2 * m + 1] for m in mask], np.float32)[None, None, :]
    rows = np.zeros((H, W, 3, 85), np.float32)
    rows[..., 0] = (sig(t[..., 0]) + xs) / np.float32(W)
    rows[..., 1] = (sig(t[..., 1]) + ys) / np.float32(H)
    rows[..., 2] = np.exp(t[..., 2], dtype=np.float32) * aw / np.float32(net_w)
    rows[..., 3] = np.exp(t[..., 3], dtype=np.float32) * ah / np.float32(net_h)
    obj = sig(t[..., 4])
    rows[..., 4] = obj
    p = obj[..., None] * sig(t[..., 5:])
    rows[..., 5:] = np.where(p > np.float32(thresh), p, np.float32(0))
    return rows.reshape(-1, 85)


KEEP_CLASSES = (0, 1, 2, 5, 7)   # person, bicycle, car, bus, truck ("motorcycle" never matches coco.names' "motorbike")


def postprocess(rows, frame_cols, frame_rows, conf_thr=0.5, nms_thr=0.4):
    """yolov3Segment::postprocess_ (yolo.cc:151-206): confidence filter, int boxes, NMSBoxes, class filter, rectCenterScale."""
    boxes, confs, cls = [], [], []
    for r in rows:
        sc = r[5:]
        c = int(np.argmax(sc)); confidence = float(sc[c])
        if confidence > np.float32(conf_thr):
            cx = int(np.float32(r[0]) * np.float32(frame_cols)); cy = int(np.float32(r[1]) * np.float32(frame_rows))
            w = int(np.float32(r[2]) * np.float32(frame_cols)); h = int(np.float32(r[3]) * np.float32(frame_rows))
            boxes.append((cx - int(w / 2), cy - int(h / 2), w, h)); confs.append(np.float32(confidence)); cls.append(c)
    order = sorted(range(len(boxes)), key=lambda i: -confs[i])      # stable

    def overlap(a, b):
        Aa, Ab = a[2] * a[3], b[2] * b[3]
        if Aa + Ab <= 2.220446049250313e-16:
            return np.float32(1)
        x1, y1 = max(a[0], b[0]), max(a[1], b[1]); x2, y2 = min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3])
        Aab = (x2 - x1) * (y2 - y1) if (x2 > x1 and y2 > y1) else 0
        return np.float32(1. - (1. - Aab / (Aa + Ab - Aab)))

    keep = []
    for i in order:
        k = True
        for j in keep:
            if not k:
                break
            k = overlap(boxes[i], boxes[j]) <= np.float32(nms_thr)
        if k:
            keep.append(i)
    out, oc, of = [], [], []
    for i in keep:
        if cls[i] in KEEP_CLASSES:
            x, y, w, h = boxes[i]
            sw, sh = -0.2 * w, 0.6 * h
            out.append((x - sw / 2.0, y - sh / 2.0, w + sw, h + sh)); oc.append(cls[i]); of.append(confs[i])
    return np.array(out, np.float64).reshape(-1, 4), np.array(oc, np.int32), np.array(of, np.float32)


def ellipse_element(ksize=31):
    """cv::getStructuringElement(MORPH_ELLIPSE, Size(k, k)) [OpenCV-recall]."""
    r = c = ksize // 2
    inv_r2 = 1.0 / (r * r)
    el = np.zeros((ksize, ksize), np.uint8)
    for i in range(ksize):
        dy = i - r
        dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
        el[i, max(c - dx, 0):min(c + dx + 1, ksize)] = 1
    return el


def segmentation_mask(rows, frame_cols, frame_rows, conf_thr=0.5, nms_thr=0.4):
    """yolov3Segment::Segmentation (yolo.cc:34-58) from region rows: (mask u8, noTarget)."""
    boxes, confs, cls = [], [], []
    for r in rows:
        sc = r[5:]
        c = int(np.argmax(sc)); confidence = float(sc[c])
        if confidence > np.float32(conf_thr):
            cx = int(np.float32(r[0]) * np.float32(frame_cols)); cy = int(np.float32(r[1]) * np.float32(frame_rows))
            w = int(np.float32(r[2]) * np.float32(frame_cols)); h = int(np.float32(r[3]) * np.float32(frame_rows))
            boxes.append((cx - int(w / 2), cy - int(h / 2), w, h)); confs.append(np.float32(confidence)); cls.append(c)
    order = sorted(range(len(boxes)), key=lambda i: -confs[i])

    def overlap(a, b):
        Aa, Ab = a[2] * a[3], b[2] * b[3]
        if Aa + Ab <= 2.220446049250313e-16:
            return np.float32(1)
        x1, y1 = max(a[0], b[0]), max(a[1], b[1]); x2, y2 = min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3])
        Aab = (x2 - x1) * (y2 - y1) if (x2 > x1 and y2 > y1) else 0
        return np.float32(1. - (1. - Aab / (Aa + Ab - Aab)))
    keep = []
    for i in order:
        if all(overlap(boxes[i], boxes[j]) <= np.float32(nms_thr) for j in keep):
            keep.append(i)
    m = np.zeros((frame_rows, frame_cols), np.uint8)
    no_target = True
    for i in keep:
        if cls[i] in KEEP_CLASSES:
            x, y, w, h = boxes[i]
            x0, x1 = max(0, x + int(w / 4)), min(x + int(3 * w / 4), frame_cols)     # C++ int division of non-negative values
            y0, y1 = max(0, y), min(y + h, frame_rows)
            if x1 > x0 and y1 > y0:
                m[y0:y1, x0:x1] = 1
            no_target = False
    if no_target:
        return np.ones((frame_rows, frame_cols), np.uint8), True
    el = ellipse_element(31)
    dil = np.zeros_like(m)
    ys, xs = np.nonzero(el)
    pad = np.pad(m, 15)
    for dy, dx in zip(ys - 15, xs - 15):
        dil |= pad[15 + dy:15 + dy + frame_rows, 15 + dx:15 + dx + frame_cols]
    return (1 - dil).astype(np.uint8), False
