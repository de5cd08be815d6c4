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


@pytest.mark.parametrize("prec", ["default", "f16", "f32w", "f32x3"])
def test_cpp_yolov3segment_matches_python_detector(gpu, pkg, fe, synth, tmp_path, prec):
    """host/yolo.h: Darknet cfg + weights files -> Segmentation_ boxes and Segmentation mask, equal to the ctypes Detector's.  The class
    computes in f32 unless told otherwise (cv::dnn's arithmetic, yolo.cc:29); the other two modes are constructor arguments."""
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
    subprocess.check_call([exe, str(cfg), str(wts), str(W), str(H), str(raw), str(out)] + ([] if prec == "default" else [str({"f16": 0, "f32w": 2, "f32x3": 3}[prec])]))
    blob = out.read_bytes()
    n = int(np.frombuffer(blob, np.int32, 1)[0])
    boxes = np.frombuffer(blob, np.float64, 4 * n, 4).reshape(n, 4)
    nt = int(np.frombuffer(blob, np.int32, 1, 4 + 32 * n)[0])
    mask = np.frombuffer(blob, np.uint8, W * H, 8 + 32 * n).reshape(H, W)
    d = yolo.Detector(layers, anchors, 640, 480, max_batch=1, precision="f32" if prec == "default" else prec)
    d.load_weights(payload)
    dev = torch.from_numpy(img[None]).cuda()
    d.forward_device(dev.data_ptr(), W, H, W * 3, W * H * 3, 1, 0.5)
    eb, _, _ = d.boxes(0, W, H)
    d_mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    ent = d.mask_device(0, W, H, d_mask.data_ptr(), W)
    torch.cuda.synchronize()
    assert n == len(eb) and n > 0 and np.array_equal(boxes, eb)
    assert nt == int(ent) and np.array_equal(mask, d_mask.cpu().numpy()) and mask.min() == 0 and mask.max() == 1


def _read_frame_dump(buf, off, fe):
    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(buf, dtype, n, off); off += a.nbytes
        return a
    N, N_ori, N_d, nb, flag, ret, ref, fid = [int(v) for v in take(np.int32, 8)]
    out = dict(N=N, N_ori=N_ori, N_d=N_d, nb=nb, flag=flag, ret=ret, ref=ref, id=fid)
    out["kp"] = take(fe.KP_DTYPE, N); out["kpUn"] = take(fe.KP_DTYPE, N); out["desc"] = take(np.uint8, 32 * N).reshape(N, 32)
    out["ur"] = take(np.float32, N); out["dep"] = take(np.float32, N)
    out["boxes"] = []
    for _ in range(nb):
        r = take(np.float64, 4); idx, st, om = [int(v) for v in take(np.int32, 3)]; vel = take(np.float64, 2)
        k = int(take(np.int32, 1)[0])
        out["boxes"].append(dict(rect=r, idx=idx, status=st, omit=om, vel=vel, kp=take(fe.KP_DTYPE, k), kpUn=take(fe.KP_DTYPE, k), desc=take(np.uint8, 32 * k).reshape(k, 32),
                                 ur=take(np.float32, k), dep=take(np.float32, k)))
    out["cell"] = take(np.int32, N)
    out["Tcw"] = take(np.float32, 16).reshape(4, 4)
    return out, off


def _yaw(cfg, px=3.0):
    a = px / float(cfg["fx"])
    V = np.eye(4, dtype=np.float32)
    V[0, 0] = np.float32(np.cos(a)); V[0, 2] = np.float32(np.sin(a)); V[2, 0] = np.float32(-np.sin(a)); V[2, 2] = np.float32(np.cos(a))
    return V


def _commit(t, F, seed):
    rng = np.random.default_rng(1000 * seed + t)
    xw = F.xw.copy(); fl = F.mp_flags.copy()
    xw += rng.normal(0, 0.01, xw.shape).astype(np.float32)
    fl[rng.random(len(fl)) < 0.15] = 0
    fl[(rng.random(len(fl)) < 0.4) & (fl != 0)] |= 2
    return xw, fl


@pytest.mark.parametrize("kind", ["stereo", "rgbd", "rgbd-tum1", "stereo-moving", "rgbd-rgba-f32", "mono"])
def test_cpp_system_track_matches_frame_oracle(gpu, fe, orc, synth, tmp_path, kind):
    """host/Frame.h: ORB_SLAM2::System::TrackStereo / TrackRGBD / TrackMonocular (C++, g++, no OpenCV) frame by frame against the frame-level
    oracle: every public Frame member this path produces, bit for bit (stereo: KITTI configs[2]; RGB-D: TUM3 configs[3] with mask + boxes).
      stereo-moving   the pose side's state through the mirror: Tracking::mState, mVelocity (the mirror predicts mVelocity * mLastFrame.mTcw as
                      Tracking.cc:982 does and hands Tcw / Twc on) and the MapPoints it commits after every frame (Tracking.cc:998-1010)
      rgbd-rgba-f32   4-channel colour + CV_32F depth still to be scaled by DepthMapFactor (Tracking.cc:187-200, 271-272), with a velocity
      mono            TrackMonocular: mpIniORBextractor (2 * nFeatures) for the first two frames (Tracking.cc:127-128, 335-338)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sd_oracle_pipeline_m", os.path.join(ROOT, "oracle", "pipeline.py"))
    P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)
    exe = str(tmp_path / "frame_mirror")
    libdir = os.path.join(ROOT, "slam-dynamic_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "slam-dynamic_amd", "host"),
                           os.path.join(ROOT, "tests/cpp/frame_mirror_main.cpp"), "-L" + libdir, "-lsd_frontend", "-Wl,-rpath," + libdir, "-o", exe])
    variant = kind
    stereo = kind in ("stereo", "stereo-moving")
    mono = kind == "mono"
    ext = kind in ("stereo-moving", "rgbd-rgba-f32")
    f32d = kind == "rgbd-rgba-f32"
    cfg = synth.KITTI_STEREO if stereo else (synth.TUM1 if kind == "rgbd-tum1" else synth.TUM3)      # TUM1: Camera.k1 != 0, mvKeysUn != mvKeys
    kind = "stereo" if stereo else ("mono" if mono else "rgbd")
    T = (8 if ext else 5) if stereo else (4 if mono else 9)
    ch = 1 if stereo else (4 if f32d else 3)
    W, H = cfg["width"], cfg["height"]
    o = P.SequenceOracle(orc, cfg, P.SENSOR_STEREO if stereo else (P.SENSOR_MONOCULAR if mono else P.SENSOR_RGBD), ini_features=2 * cfg["n_features"] if mono else 0)
    blob, ref, poses = [], [], []
    V = _yaw(cfg)
    Tcw = np.eye(4, dtype=np.float32)
    for t in range(T):
        ts = t / cfg["fps"]
        rows = synth.boxes_for_frame(71, t, cfg)
        boxes = synth.rows_to_rects(rows) if (t != 1 and not mono) else None          # frame 1 goes through the overload without boxes
        if stereo:
            a, b, _ = synth.stereo_frame_dyn(71, t, cfg); extra = b""
        elif mono:
            a = synth.rgbd_frame_dyn(71, t, cfg)[0]; b = None; extra = b""
        else:
            a, b, _ = synth.rgbd_frame_dyn(71, t, cfg); extra = synth.mask_from_boxes(rows, W, H).tobytes()
            if f32d:
                a = np.ascontiguousarray(np.concatenate([a, np.full(a.shape[:2] + (1,), 200, np.uint8)], -1))
                b = b.astype(np.float32)                               # CV_32F, still in DepthMapFactor units: convertTo scales it
        head = b""
        state = None
        if ext:                                                        # the pose side: NOT_INITIALIZED, then OK; a velocity from the third frame on
            mstate, hv = (1, 0) if t == 0 else (2, 1 if t >= 2 else 0)
            if variant == "stereo-moving" and t == 5:
                mstate = 3                                             # LOST on frame 5: no TrackHomo
            if hv:
                Tcw = P.pose_mul(V, Tcw)                               # mCurrentFrame.SetPose(mVelocity*mLastFrame.mTcw) (Tracking.cc:982)
            head = np.array([mstate, hv], np.int32).tobytes() + V.tobytes()
            state = (1 if mstate in (2, 3) else 0) | (2 if (mstate == 2 and hv) else 0)
        F = o.track(a, b, boxes, ts, Tcw=Tcw if ext else None, Twc=P.pose_inverse(Tcw) if ext else None, state=state)
        tail = b""
        if ext:
            if variant == "stereo-moving":
                xw, fl = _commit(t, F, 71)
                o.set_mappoints(xw, fl)
                tail = np.array([len(fl)], np.int32).tobytes() + xw.tobytes() + fl.tobytes()
            else:
                tail = np.array([-1], np.int32).tobytes()
        blob.append(head + np.array([ts], np.float64).tobytes() + np.array([-1 if boxes is None else len(boxes)], np.int32).tobytes() +
                    (b"" if boxes is None else boxes.tobytes()) + a.tobytes() + (b"" if b is None else b.tobytes()) + extra + tail)
        ref.append(F); poses.append(Tcw.copy())
    inp = tmp_path / "in.bin"; out = tmp_path / "out.bin"
    inp.write_bytes(b"".join(blob))
    subprocess.check_call([exe, kind, str(W), str(H), str(ch), str(T), str(inp), str(out), repr(float(np.float32(cfg["fx"]))), repr(float(np.float32(cfg["fy"]))),
                           repr(float(np.float32(cfg["cx"]))), repr(float(np.float32(cfg["cy"]))), repr(float(np.float32(cfg["bf"]))), str(cfg["fps"]),
                           str(cfg.get("depth_map_factor", 1.0)), str(cfg["n_features"]), str(cfg["ini_th_fast"])] +
                          [repr(float(np.float32(cfg.get(k, 0.0)))) for k in ("k1", "k2", "p1", "p2", "k3")] + [str(int(ext)), str(int(f32d))])
    buf = out.read_bytes()
    off = 0
    ran = 0
    for t, F in enumerate(ref):
        g, off = _read_frame_dump(buf, off, fe)
        tag = "%s frame %d" % (kind, t)
        assert (g["N"], g["N_ori"], g["N_d"], g["nb"], g["id"]) == (F.N, F.N_s, F.N_d, len(F.objects), F.mnId), tag
        assert (g["flag"], g["ref"]) == (F.track_flag, F.ref_id) and g["ret"] == (F.separate_ret or 0), tag
        assert g["kp"].tobytes() == F.kp.tobytes() and g["kpUn"].tobytes() == F.kpUn.tobytes() and np.array_equal(g["desc"], F.desc), tag + ": mvKeys / mvKeysUn / mDescriptors"
        assert np.array_equal(g["ur"].view(np.uint32), F.ur.view(np.uint32)) and np.array_equal(g["dep"].view(np.uint32), F.dep.view(np.uint32)), tag
        assert np.array_equal(g["cell"], F.cells), tag + ": mGrid"
        if ext:
            assert g["Tcw"].tobytes() == poses[t].tobytes(), tag + ": the pose prior mVelocity * mLastFrame.mTcw"
        for j, bx in enumerate(g["boxes"]):
            assert np.array_equal(bx["rect"], F.objects[j]) and (bx["idx"], bx["status"], bx["omit"]) == (F.box_idx[j], F.box_status[j], F.omit[j]), tag
            assert np.array_equal(bx["vel"], F.velocity[j]), tag
            it = F.boxItems[F.boxStart[j]:F.boxStart[j + 1]]
            assert bx["kp"].tobytes() == F.dyn_kp[it].tobytes() and np.array_equal(bx["desc"], F.dyn_desc[it]), tag + ": mvdynKeys / mdynDescriptors"
            assert bx["kpUn"].tobytes() == F.dyn_kpUn[it].tobytes(), tag + ": mvdynKeysUn"
            assert np.array_equal(bx["ur"].view(np.uint32), F.dyn_ur[it].view(np.uint32)) and np.array_equal(bx["dep"].view(np.uint32), F.dyn_dep[it].view(np.uint32)), tag
        ran += F.track_flag != 0
    assert off == len(buf) and (ran >= 1 or mono), "the dynamic block must have run at least once"
    if variant == "stereo-moving":
        assert ref[5].track_flag == 0 and ref[6].track_flag != 0, "LOST on frame 5 (no TrackHomo), tracking again on frame 6"
    if mono:
        assert ref[0].N > 1.5 * cfg["n_features"] and ref[1].N > 1.5 * cfg["n_features"] and ref[2].N < 1.2 * cfg["n_features"], "mpIniORBextractor on the first two frames"
