"""The C++ class-API mirror (slam-dynamic_amd/host/ORBextractor.h) built with g++ against the C ABI and run on the GPU."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_orbextractor_matches_oracle(gpu, fe, orc, synth, tmp_path):
    exe = str(tmp_path / "host_mirror")
    libdir = os.path.join(ROOT, "slam-dynamic_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "slam-dynamic_amd", "host"), os.path.join(ROOT, "tests/cpp/host_mirror_main.cpp"),
                           "-L" + libdir, "-lsd_frontend", "-Wl,-rpath," + libdir, "-o", exe])
    img = synth.random_image(640, 480, 31)
    raw = tmp_path / "img.raw"; out = tmp_path / "out.bin"
    raw.write_bytes(img.tobytes())
    subprocess.check_call([exe, "640", "480", str(raw), str(out), "1000", "20", "7"])
    blob = out.read_bytes()
    n = int(np.frombuffer(blob, np.int32, 1)[0])
    kp = np.frombuffer(blob, fe.KP_DTYPE, n, 4)
    desc = np.frombuffer(blob, np.uint8, n * 32, 4 + 28 * n).reshape(n, 32)
    o = orc.Extractor(1000, 1.2, 8, 20, 7)
    rk, rd = o(img)
    assert n == len(rk) and kp.tobytes() == rk.tobytes() and np.array_equal(desc, rd)
    off = 4 + 60 * n
    w1, h1 = np.frombuffer(blob, np.int32, 2, off)
    plane = np.frombuffer(blob, np.uint8, (w1 + 38) * (h1 + 38), off + 8).reshape(h1 + 38, w1 + 38)
    assert np.array_equal(plane, o.pyramid(1))


def test_cpp_yolov3segment_matches_python_detector(gpu, pkg, fe, synth, tmp_path):
    """host/yolo.h: Darknet cfg + weights files -> Segmentation_ boxes and Segmentation mask, equal to the ctypes Detector's."""
    import torch
    yolo = pkg.yolo
    exe = str(tmp_path / "yolo_mirror")
    libdir = os.path.join(ROOT, "slam-dynamic_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "slam-dynamic_amd", "host"), os.path.join(ROOT, "tests/cpp/yolo_mirror_main.cpp"),
                           "-L" + libdir, "-lsd_frontend", "-Wl,-rpath," + libdir, "-o", exe])
    layers, anchors = yolo.v3_layers()
    payload, _ = yolo.synth_weights(layers, seed=3)
    cfg = tmp_path / "yolov3.cfg"; wts = tmp_path / "yolov3.weights"
    yolo.write_cfg(cfg, layers, anchors)
    yolo.write_darknet_weights(wts, payload)
    pl, pa, pc = yolo.parse_cfg(cfg)                                   # the Python parser reads the same file back
    assert np.array_equal(pl, layers) and np.array_equal(pa, anchors) and pc == 80
    c = synth.KITTI03_RGBD
    img = np.ascontiguousarray(synth.rgbd_frame(6, 0, c)[0][:, :, ::-1])
    H, W = img.shape[:2]
    raw = tmp_path / "img.raw"; out = tmp_path / "out.bin"
    raw.write_bytes(img.tobytes())
    subprocess.check_call([exe, str(cfg), str(wts), str(W), str(H), str(raw), str(out)])
    blob = out.read_bytes()
    n = int(np.frombuffer(blob, np.int32, 1)[0])
    boxes = np.frombuffer(blob, np.float64, 4 * n, 4).reshape(n, 4)
    nt = int(np.frombuffer(blob, np.int32, 1, 4 + 32 * n)[0])
    mask = np.frombuffer(blob, np.uint8, W * H, 8 + 32 * n).reshape(H, W)
    d = yolo.Detector(layers, anchors, 640, 480, max_batch=1)
    d.load_weights(payload)
    dev = torch.from_numpy(img[None]).cuda()
    d.forward_device(dev.data_ptr(), W, H, W * 3, W * H * 3, 1, 0.5)
    eb, _, _ = d.boxes(0, W, H)
    d_mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    ent = d.mask_device(0, W, H, d_mask.data_ptr(), W)
    torch.cuda.synchronize()
    assert n == len(eb) and n > 0 and np.array_equal(boxes, eb)
    assert nt == int(ent) and np.array_equal(mask, d_mask.cpu().numpy()) and mask.min() == 0 and mask.max() == 1
