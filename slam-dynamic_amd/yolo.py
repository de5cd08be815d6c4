"""Detector plumbing: Darknet cfg parsing, synthetic Darknet weight payloads, ctypes wrapper of sd_yolo_*.

`yolov3.weights` is a URL download the reference never shipped (README.md:41) and there is no network, so
parity/benchmarks use seeded synthetic weights of the right shapes in the .weights payload layout."""
import ctypes as C

import numpy as np

from . import frontend as fe

CONV, SHORTCUT, ROUTE, UPSAMPLE, YOLO = 0, 1, 2, 3, 4

LAYER_DTYPE = np.dtype([("type", "<i4"), ("filters", "<i4"), ("size", "<i4"), ("stride", "<i4"), ("batch_normalize", "<i4"),
                        ("leaky", "<i4"), ("from", "<i4", (2,)), ("nfrom", "<i4"), ("mask", "<i4", (3,))])


def parse_cfg(path):
    """Parse a Darknet cfg into (layers, anchors, classes).  Only the 5 layer types of yolov3.cfg are accepted."""
    sections, cur = [], None
    for line in open(path):
        line = line.split("#")[0].strip()
        if not line:
            continue
        if line.startswith("["):
            cur = {"_type": line.strip("[]")}
            sections.append(cur)
        else:
            k, v = line.split("=", 1)
            cur[k.strip()] = v.strip()
    layers = np.zeros(len(sections) - 1, LAYER_DTYPE)
    anchors, classes = None, 80
    for i, s in enumerate(sections[1:]):
        t = s["_type"]
        L = layers[i]
        if t == "convolutional":
            L["type"] = CONV; L["filters"] = int(s["filters"]); L["size"] = int(s["size"]); L["stride"] = int(s["stride"])
            L["batch_normalize"] = int(s.get("batch_normalize", 0)); L["leaky"] = int(s["activation"] == "leaky")
            assert int(s.get("pad", 0)) == 1
        elif t == "shortcut":
            L["type"] = SHORTCUT; L["from"][0] = int(s["from"]); L["nfrom"] = 1
        elif t == "route":
            ls = [int(v) for v in s["layers"].split(",")]
            L["type"] = ROUTE; L["nfrom"] = len(ls); L["from"][:len(ls)] = ls
        elif t == "upsample":
            L["type"] = UPSAMPLE; L["stride"] = int(s["stride"])
        elif t == "yolo":
            L["type"] = YOLO; L["mask"] = [int(v) for v in s["mask"].split(",")]
            anchors = np.array([float(v) for v in s["anchors"].split(",")], np.float32); classes = int(s["classes"])
        else:
            raise ValueError("unsupported section " + t)
    return layers, anchors, classes


def write_cfg(path, layers, anchors, classes=80, net_w=640, net_h=480):
    """A Darknet .cfg with the sections of src/yolo/yolov3.cfg for the given layer list (what parse_cfg / host/yolo.h read)."""
    with open(path, "w") as f:
        f.write("[net]\n# Testing\nbatch=1\nsubdivisions=1\nwidth=%d\nheight=%d\nchannels=3\n\n" % (net_w, net_h))
        for L in layers:
            t = int(L["type"])
            if t == CONV:
                f.write("[convolutional]\n")
                if L["batch_normalize"]:
                    f.write("batch_normalize=1\n")
                f.write("filters=%d\nsize=%d\nstride=%d\npad=1\nactivation=%s\n\n" % (L["filters"], L["size"], L["stride"], "leaky" if L["leaky"] else "linear"))
            elif t == SHORTCUT:
                f.write("[shortcut]\nfrom=%d\nactivation=linear\n\n" % L["from"][0])
            elif t == ROUTE:
                f.write("[route]\nlayers = %s\n\n" % ", ".join(str(int(v)) for v in L["from"][:L["nfrom"]]))
            elif t == UPSAMPLE:
                f.write("[upsample]\nstride=%d\n\n" % L["stride"])
            elif t == YOLO:
                f.write("[yolo]\nmask = %s\nanchors = %s\nclasses=%d\nnum=9\njitter=.3\nignore_thresh = .7\ntruth_thresh = 1\nrandom=1\n\n"
                        % (",".join(str(int(v)) for v in L["mask"]), ",  ".join("%d,%d" % (anchors[2 * k], anchors[2 * k + 1]) for k in range(9)), classes))


def write_darknet_weights(path, payload, major=0, minor=2, revision=0, seen=32013312):
    """yolov3.weights layout: int32 major, minor, revision, int64 seen (major*10 + minor >= 2), then the floats."""
    with open(path, "wb") as f:
        f.write(np.array([major, minor, revision], np.int32).tobytes())
        f.write(np.array([seen], np.int64 if major * 10 + minor >= 2 else np.int32).tobytes())
        f.write(np.ascontiguousarray(payload, np.float32).tobytes())


def v3_layers():
    """The built-in YOLOv3 layer list of the C ABI (sd_yolo_v3_layers)."""
    L = fe.lib()
    n = C.c_int()
    layers = np.zeros(256, LAYER_DTYPE); anchors = np.zeros(18, np.float32)
    L.sd_yolo_v3_layers.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]
    fe.check(L.sd_yolo_v3_layers(fe._p(layers), 256, C.byref(n), fe._p(anchors)))
    return layers[:n.value].copy(), anchors


def conv_inputs(layers):
    """Input channel count of every convolution (3 for the first)."""
    C_, out_c, cins = 3, [], {}
    for i, l in enumerate(layers):
        t = l["type"]
        if t == CONV:
            cins[i] = C_; C_ = int(l["filters"])
        elif t == ROUTE:
            C_ = sum(out_c[i + f if f < 0 else f] for f in l["from"][:l["nfrom"]])
        out_c.append(C_)
    return cins


def synth_weights(layers, seed=7):
    """Seeded Darknet-payload weights: He-initialised filters, mild batch-norm statistics; residual branches are damped
    (gamma 0.3) so that 23 shortcuts do not blow up f16.  Returns (payload f32, per-conv dict for a torch reference)."""
    rng = np.random.default_rng(seed)
    cins = conv_inputs(layers)
    parts, per = [], {}
    for i, l in enumerate(layers):
        if l["type"] != CONV:
            continue
        F, k, cin = int(l["filters"]), int(l["size"]), cins[i]
        fan_in = cin * k * k
        w = rng.normal(0, np.sqrt(2.0 / fan_in / (1 + 0.01)), (F, cin, k, k)).astype(np.float32)
        damp = 0.3 if (i + 1 < len(layers) and layers[i + 1]["type"] == SHORTCUT) else 1.0
        if l["batch_normalize"]:
            beta = rng.normal(0, 0.05, F).astype(np.float32)
            gamma = (rng.uniform(0.9, 1.1, F) * damp).astype(np.float32)
            mean = rng.normal(0, 0.05, F).astype(np.float32)
            var = rng.uniform(0.8, 1.2, F).astype(np.float32)
            parts += [beta, gamma, mean, var, w.reshape(-1)]
            per[i] = dict(w=w, beta=beta, gamma=gamma, mean=mean, var=var)
        else:
            bias = rng.normal(0, 0.2, F).astype(np.float32)
            if F == 255:
                w *= 0.12                  # head inputs have std ~9 with these weights: bring the logits back to O(1)
                bias[4::85] = -1.0         # objectness prior: on the order of a thousand rows pass the 0.5 threshold
                for c in (0, 2, 5):        # person / car / bus slightly favoured so that the class filter keeps some boxes
                    bias[5 + c::85] += 1.0
            parts += [bias, w.reshape(-1)]
            per[i] = dict(w=w, bias=bias)
    return np.concatenate(parts).astype(np.float32), per


class Detector:
    """yolov3Segment (include/yolo.h:22-48) on the GPU."""

    def __init__(self, layers=None, anchors=None, net_w=640, net_h=480, max_batch=1, precision="f16"):
        L = fe.lib()
        if layers is None:
            layers, anchors = v3_layers()
        self.layers, self.anchors = np.ascontiguousarray(layers), np.ascontiguousarray(anchors, np.float32)
        self.net_w, self.net_h = net_w, net_h
        self.h = C.c_void_p()
        vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        L.sd_yolo_create_prec.argtypes = [C.POINTER(vp), vp, i, vp, i, i, i, i, i]
        self.precision = precision
        L.sd_yolo_destroy.argtypes = [vp]
        L.sd_yolo_weight_count.argtypes = [vp, C.POINTER(sz)]
        L.sd_yolo_load_darknet_weights.argtypes = [vp, vp, sz]
        L.sd_yolo_layer_shape.argtypes = [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
        L.sd_yolo_flops.argtypes = [vp, C.POINTER(C.c_double)]
        L.sd_yolo_mfma_flops.argtypes = [vp, C.POINTER(C.c_double)]
        L.sd_yolo_forward_device.argtypes = [vp, vp, i, i, sz, sz, i, f, vp]
        L.sd_yolo_download_layer.argtypes = [vp, i, i, vp]
        L.sd_yolo_download_region.argtypes = [vp, vp, C.POINTER(i)]
        L.sd_yolo_boxes.argtypes = [vp, i, i, i, f, f, vp, vp, vp, i, C.POINTER(i)]
        L.sd_yolo_mask_device.argtypes = [vp, i, i, i, f, f, vp, sz, C.POINTER(i), vp]
        fe.check(L.sd_yolo_create_prec(C.byref(self.h), fe._p(self.layers), len(self.layers), fe._p(self.anchors), 80, net_w, net_h, max_batch,
                                       {"f16": 0, "f32": 1, "f32w": 2, "f32x3": 3}[precision]))

    def close(self):
        if self.h:
            fe.lib().sd_yolo_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def weight_count(self):
        n = C.c_size_t(); fe.check(fe.lib().sd_yolo_weight_count(self.h, C.byref(n))); return n.value

    def load_weights(self, payload):
        p = np.ascontiguousarray(payload, np.float32)
        fe.check(fe.lib().sd_yolo_load_darknet_weights(self.h, fe._p(p), len(p)))

    def layer_shape(self, layer):
        h, w, c = C.c_int(), C.c_int(), C.c_int()
        fe.check(fe.lib().sd_yolo_layer_shape(self.h, layer, C.byref(h), C.byref(w), C.byref(c)))
        return h.value, w.value, c.value

    def flops(self):
        d = C.c_double(); fe.check(fe.lib().sd_yolo_flops(self.h, C.byref(d))); return d.value

    def mfma_flops(self):
        """MFMA FLOPs per image as the mode executes them (== flops() except for "f32w", whose Winograd layers run 16 / 36 of the multiplies)."""
        d = C.c_double(); fe.check(fe.lib().sd_yolo_mfma_flops(self.h, C.byref(d))); return d.value

    def mfma_flops_bf16(self):
        d = C.c_double(); fe.lib().sd_yolo_mfma_flops_bf16.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        fe.check(fe.lib().sd_yolo_mfma_flops_bf16(self.h, C.byref(d))); return d.value

    def winograd_layers(self):
        n = C.c_int(); fe.lib().sd_yolo_winograd_layers.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        fe.check(fe.lib().sd_yolo_winograd_layers(self.h, C.byref(n))); return n.value

    def forward_device(self, d_bgr_ptr, width, height, stride, pitch, n, conf=0.5, stream=None):
        fe.check(fe.lib().sd_yolo_forward_device(self.h, C.c_void_p(d_bgr_ptr), width, height, stride, pitch, n, conf, C.c_void_p(stream or 0)))

    def layer_output(self, layer, image=0):
        h, w, c = self.layer_shape(layer)
        out = np.zeros((h, w, c), np.float32 if self.precision in ("f32", "f32w", "f32x3") else np.float16)
        fe.check(fe.lib().sd_yolo_download_layer(self.h, layer, image, fe._p(out)))
        return out

    def region_rows(self):
        n = C.c_int()
        rows = np.zeros((4 * 18900, 85), np.float32)
        fe.check(fe.lib().sd_yolo_download_region(self.h, fe._p(rows), C.byref(n)))
        return rows[:n.value].copy()

    def boxes(self, image, frame_cols, frame_rows, conf=0.5, nms=0.4, cap=1024):
        b = np.zeros((cap, 4), np.float64); cid = np.zeros(cap, np.int32); cf = np.zeros(cap, np.float32)
        n = C.c_int()
        fe.check(fe.lib().sd_yolo_boxes(self.h, image, frame_cols, frame_rows, conf, nms, fe._p(b), fe._p(cid), fe._p(cf), cap, C.byref(n)))
        return b[:n.value].copy(), cid[:n.value].copy(), cf[:n.value].copy()

    def boxes_batch(self, n_images, frame_cols, frame_rows, conf=0.5, nms=0.4, stream=None):
        """postprocess_ of every image of the last forward pass, NMS on the device, one download -> list of (boxes, cls, conf)."""
        M = fe.MAXB
        b = np.zeros((n_images, M, 4), np.float64); cid = np.zeros((n_images, M), np.int32); cf = np.zeros((n_images, M), np.float32)
        nb = np.zeros(n_images, np.int32)
        fe.check(fe.lib().sd_yolo_boxes_batch(self.h, n_images, frame_cols, frame_rows, C.c_float(conf), C.c_float(nms), fe._p(b), fe._p(cid),
                                              fe._p(cf), fe._p(nb), C.c_void_p(stream or 0)))
        return [(b[i, :nb[i]].copy(), cid[i, :nb[i]].copy(), cf[i, :nb[i]].copy()) for i in range(n_images)]

    def set_overlap(self, on=True):
        """sd_yolo_set_overlap: blobFromImage / region decodes on internal streams around the convolution stream (f32-class modes)."""
        L = fe.lib()
        L.sd_yolo_set_overlap.argtypes = [C.c_void_p, C.c_int]
        fe.check(L.sd_yolo_set_overlap(self.h, int(bool(on))))

    def boxes_device(self, n_images, frame_cols, frame_rows, d_boxes, d_cls, d_conf, d_n, conf=0.5, nms=0.4, stream=None):
        """The same post-processing into caller-owned DEVICE buffers ([n][MAXB][4] f64, [n][MAXB] i32, [n][MAXB] f32, [n] i32; MAXB = SD_MAX_BOXES), no synchronisation."""
        L = fe.lib()
        L.sd_yolo_boxes_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        fe.check(L.sd_yolo_boxes_device(self.h, n_images, frame_cols, frame_rows, conf, nms, C.c_void_p(d_boxes), C.c_void_p(d_cls), C.c_void_p(d_conf),
                                        C.c_void_p(d_n), C.c_void_p(stream or 0)))

    def mask_device(self, image, frame_cols, frame_rows, d_mask_ptr, stride, conf=0.5, nms=0.4, stream=None):
        nt = C.c_int()
        fe.check(fe.lib().sd_yolo_mask_device(self.h, image, frame_cols, frame_rows, conf, nms, C.c_void_p(d_mask_ptr), stride,
                                              C.byref(nt), C.c_void_p(stream or 0)))
        return bool(nt.value)
