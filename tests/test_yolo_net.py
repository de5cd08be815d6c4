"""Detector network description (no GPU): the built-in YOLOv3 layer list equals the reference's cfg file."""
import os

import numpy as np

CFG = "/root/reference/src/yolo/yolov3.cfg"


def test_builtin_v3_layers_match_reference_cfg(pkg):
    layers, anchors = pkg.yolo.v3_layers()
    assert len(layers) == 107 and (layers["type"] == pkg.yolo.CONV).sum() == 75
    assert (layers["type"] == pkg.yolo.SHORTCUT).sum() == 23 and (layers["type"] == pkg.yolo.YOLO).sum() == 3
    assert anchors.tolist() == [10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326]
    if not os.path.exists(CFG):
        return          # GPU box: the reference tree is absent; the structural asserts above stand in
    ref, ranchors, classes = pkg.yolo.parse_cfg(CFG)
    assert classes == 80 and np.array_equal(ranchors, anchors)
    assert len(ref) == len(layers)
    for i, (a, b) in enumerate(zip(layers, ref)):
        assert a["type"] == b["type"], i
        if a["type"] == pkg.yolo.CONV:
            assert all(a[k] == b[k] for k in ("filters", "size", "stride", "batch_normalize", "leaky")), i
        elif a["type"] in (pkg.yolo.SHORTCUT, pkg.yolo.ROUTE):
            assert a["nfrom"] == b["nfrom"] and list(a["from"][:a["nfrom"]]) == list(b["from"][:b["nfrom"]]), i
        elif a["type"] == pkg.yolo.YOLO:
            assert list(a["mask"]) == list(b["mask"]), i


def test_weight_payload_size_and_flops(pkg):
    layers, _ = pkg.yolo.v3_layers()
    payload, per = pkg.yolo.synth_weights(layers, seed=1)
    assert len(payload) == 62001757          # 62,001,757 parameters
    assert len(per) == 75
