#!/usr/bin/env python3
"""bench.py -- tracking front-end frames/s (extract + match + dynamic cull) on MI355X, through the Frame-level C ABI (sd_tracker).

One "step" = one frame of every lane (a lane = one independent camera stream; the cull's recurrence -- boxTrack, the
reference frame 0.2-0.3 s back, box_status of the last frame -- runs along a lane, the batch runs across lanes).  Inputs
are resident in HBM before the timed region.  Default workload = BASELINE.json configs[2], the configuration the metric is
quoted on:

  stereo-yolo   KITTI stereo 1241x376 colour pairs, 2000 features: YOLOv3 @640x480 on the left image (own stream, one
                step ahead) -> Segmentation_ boxes (device NMS) -> System::TrackStereo(imLeft, imRight, boxes, t) =
                cvtColor x2 (fused) + ORB extract x2 + ComputeStereoMatches + boxTrack + firstSeparate + TrackHomo
                (SearchByProjection vs the queued frame > 0.2 s back, H / F fit) + Separate + UpdateFrame + grid +
                SearchByProjection vs mLastFrame + queue push
  stereo        the same without detector and boxes (Frame.cc:66-126)
  rgbd          BASELINE configs[1]: KITTI-03 RGB-D 1241x376, TrackRGBD(im, depth, t), no boxes / mask (Frame.cc:240-294)
  rgbd-cull     RGB-D 1241x376 with 3 given boxes per frame (rgbd_my.cc's yolov5 box files): the cull without a detector
  rgbd-bow      rgbd + Frame::ComputeBoW (Frame.cc:803-810) of the current frame and of the frame 0.2 s back, then
                ORBmatcher::SearchByBoW (ORBmatcher.cc:159-288) between the two: the timed consumer of the vocabulary that
                rank 0 broadcasts over RCCL at start-up
  tum-mask      BASELINE configs[3]: TUM3 640x480 RGB-D, DepthMapFactor 5000, mask + boxes, cull + dense back-projection
                of the unmasked pixels (pointcloudmapping.cc:59-103) -- the consumer of the semantic mask
  kitti-batch   BASELINE configs[4]: 11 sequences x 256 colour stereo frames, the stereo-yolo chain on every frame, FRAMES sharded over the
                ranks: the history-free half of a frame (detector, extraction, stereo matching: > 95 % of its time) runs on whichever rank
                the frame is dealt to, one all-to-all per time block hands the results to the rank that owns the sequence, where the
                recurrence runs frame by frame (a sequence cannot be cut: boxTrack's ids depend on its whole history); strong scaling
The non-default single-GPU workloads are also run (short) by the default invocation and reported under "extra".

Multi-GPU: one process per GPU.  `python bench.py --gpus N` without a torch.distributed environment starts the N ranks
itself (child processes, before anything touches the GPU); under `python -m torch.distributed.run` it joins the given
world.  Lanes are independent: no data-path collective (kitti-batch: one all-to-all of frame records per time block).  Start-up: RCCL broadcast of the packed ORB vocabulary from rank 0.
Per step: asynchronous gather (to rank 0) of the per-frame result records (N, N_s, keypoints, descriptors, uRight, depth,
boxes / box_idx / box_status), overlapped with the next step.  value = frames of ALL ranks / max rank time.

Prints ONE compact JSON line (< 4 KB) on rank 0; the full record (per-kernel tables, the extras' rooflines) goes to bench_detail.json.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming)
MFMA_PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3, "f32w": 157.3, "f32x3": 2500.0}   # dense peaks, MI355X_MICROARCH.md
# the detector-less workloads first: after a minute of detector passes the chip runs its vector-bound kernels ~9 % slower for a while (measured: `stereo` 71.6 k
# frames/s after a detector-less workload, 69.3 k right after the headline, 65.2 k after three detector workloads; not a queue or memory effect)
AUTO_EXTRAS = ("stereo", "rgbd", "tum-mask", "kitti-batch", "stereo-yolo-f32x3", "stereo-yolo-f32w")
WORKLOADS = ["stereo-yolo", "stereo-yolo-f32w", "stereo-yolo-f32x3", "stereo-yolo-f16", "stereo", "rgbd", "rgbd-cull", "rgbd-bow", "tum-mask", "kitti-batch"]


# --------------------------------------------------------------------------- host logic (also used by CPU tests)
def shard_sequences(n_sequences, lengths, world):
    """Longest-first greedy assignment of whole sequences to ranks."""
    order = sorted(range(n_sequences), key=lambda s: (-lengths[s], s))
    load = [0] * world
    owner = [None] * n_sequences
    for s in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[s] = r
        load[r] += lengths[s]
    return owner


def level_sizes(width, height, inv_scale):
    return [(int(np.rint(np.float32(width) * s)), int(np.rint(np.float32(height) * s))) for s in inv_scale]


def algorithmic_bytes(width, height, inv_scale, n_features, channels=1):
    """Compulsory HBM bytes per image and per kernel: each plane read/written once."""
    sizes = level_sizes(width, height, inv_scale)
    interior = [w * h for (w, h) in sizes]
    padded = [(w + 38) * (h + 38) for (w, h) in sizes]
    b = {
        "k_pyr_level0": channels * width * height + padded[0],
        "k_pyr_level": sum(interior[:-1]) + sum(padded[1:]),
        "k_fast_cells": sum(interior),
        "k_blur": 2 * sum(interior),
        "k_orient": n_features * 749,
        "k_describe": n_features * 512 + n_features * 32,
    }
    b["k_stereo_match"] = 2 * n_features * 32 + n_features * 11 * (11 + 21)      # per FRAME: both descriptor sets + the SAD windows
    b["image_total"] = (channels * width * height + sum(padded) + sum(interior[:-1]) + 3 * sum(interior) + n_features * 749 +
                        n_features * 512 + n_features * 60)
    return b


KERNEL_SYMBOL = {"k_fast_cells": "k_fast_cells_staged", "k_blur": "k_blur_wide", "k_pyr_level": "k_pyr_level_tiles", "k_pyr_level0": "k_pyr_level0_rgb"}


def load_pmc_traffic(workload):
    """profiles/pmc_traffic.json: {workload: {kernel symbol: {hbm_bytes_per_launch, batch_images, profile}}} written by tools/summarize_prof.py
    from the separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of THIS workload."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return j.get(workload, {})
    except Exception:
        return {}


def kernel_roofline_table(kt, prof_steps, alg, n_img, n_frames, pmc):
    """Per-kernel {alg_bytes, ms, GB/s, frac, traffic} of the HBM-bound front-end kernels from the in-library hipEvent times (untimed pass).
    One "launch" of k_pyr_level is the 7 dependent level launches of a step taken together."""
    out = {}
    for k, (ms, launches) in kt.items():
        if launches <= 0 or alg.get(k, 0) <= 0:
            continue
        units = n_frames if k == "k_stereo_match" else n_img
        ms_step = ms / prof_steps
        per_step = alg[k] * units
        gbs = per_step / (ms_step * 1e-3) / 1e9
        e = pmc.get(k) or pmc.get(KERNEL_SYMBOL.get(k, k))
        traffic, prof = None, None
        if e and e.get("batch_images") == n_img:
            traffic = int(e["hbm_bytes_per_launch"] * (launches / prof_steps if k == "k_pyr_level" else 1))
            prof = e.get("profile")
        out[k] = {"alg_bytes_per_step": int(per_step), "ms_per_step": round(ms_step, 4), "launches_per_step": round(launches / prof_steps, 2),
                  "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic_bytes_per_step": traffic,
                  "traffic_over_alg": round(traffic / per_step, 2) if traffic else None, "traffic_profile": prof}
    return out


def max_over_ranks(dist, seconds, device):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(dist, local, world):
    """all_gather of one rank's fixed-stride result block (uint8 tensor) -> [world, nbytes] (kept for tests / small runs)."""
    import torch
    if dist is None or world == 1:
        return local.unsqueeze(0)
    flat = local.reshape(-1)
    out = torch.empty((world * flat.numel(),), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, flat)
    return out.view((world,) + tuple(local.shape))


RECORD_FIELDS = ("count", "fb", "kp", "desc", "uright", "depth")      # N, boxes / box_status / N_s / N_d, kp, desc, uRight, depth


def record_layout(cap, fb_bytes):
    """Byte layout of one frame's result record (fixed stride): name -> (offset, bytes)."""
    sizes = {"count": 4, "fb": fb_bytes, "kp": cap * 28, "desc": cap * 32, "uright": cap * 4, "depth": cap * 4}
    off, out = 0, {}
    for k in RECORD_FIELDS:
        out[k] = (off, sizes[k]); off += (sizes[k] + 15) // 16 * 16
    out["_stride"] = off
    return out


def decode_record(rec_u8, layout, cap):
    """rec_u8: one frame's record (numpy uint8) -> dict(N, N_s, N_d, n_boxes, box_idx, box_status, kp, desc, uright, depth, readmitted)."""
    def part(k):
        o, n = layout[k]
        return rec_u8[o:o + n]
    N = int(part("count").view(np.int32)[0])
    fb = part("fb")
    head = fb[:16].view(np.int32)                     # SdFrameBoxes: nb, nAll, nOri, nDyn
    nb, n_s = int(head[0]), int(head[2])
    M = (len(fb) - 24) // 48                         # SD_MAX_BOXES from sizeof(sd_frame_boxes) = 16 + M * 32 + 3 * M * 4 + (M + 1) * 4 + 4
    base = 16 + M * 4 * 8
    box_idx = fb[base:base + 4 * M].view(np.int32)[:nb].copy()
    box_status = fb[base + 4 * M:base + 8 * M].view(np.int32)[:nb].copy()
    kp = part("kp").view(np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]))[:N]
    readmitted = np.arange(N) >= n_s                 # dyn_mask of SURVEY 8e: keypoints UpdateFrame appended behind the static ones
    return dict(N=N, N_s=n_s, N_d=int(head[3]), n_boxes=nb, box_idx=box_idx, box_status=box_status, kp=kp.copy(),
                desc=part("desc").reshape(cap, 32)[:N].copy(), uright=part("uright").view(np.float32)[:N].copy(),
                depth=part("depth").view(np.float32)[:N].copy(), readmitted=readmitted)


class ResultGather:
    """Per-rank result gather to rank 0 (north star: "per-rank result gather"), overlapped with the next step.

    The records are first packed into a staging tensor on the compute stream, then `dist.gather(..., dst=0, async_op=True)`
    moves them while the next step computes.  A gather to one root uses the 7 direct xGMI links into rank 0 in parallel; an
    all_gather would move 7x the bytes into every rank for no consumer."""

    def __init__(self, dist, world, rank, nbytes, device):
        import torch
        self.dist, self.world, self.rank = dist, world, rank
        self.stage = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        self.recv = [torch.empty((nbytes,), dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
        self.work = None

    def submit(self, parts):
        """parts: list of (dst_view_of_stage, src_tensor) pairs or plain tensors packed back to back."""
        if self.work is not None:
            self.work.wait()                     # previous gather must have drained the staging tensor
        off = 0
        for p in parts:
            if isinstance(p, tuple):
                p[0].copy_(p[1], non_blocking=True)
            else:
                n = p.numel()
                self.stage[off:off + n].copy_(p, non_blocking=True)
                off += n
        self.work = self.dist.gather(self.stage, gather_list=self.recv, dst=0, async_op=True)

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None


def spawn_ranks(n, argv, script=None):
    """`python bench.py --gpus N` outside torch.distributed: start N ranks as child processes (nothing in THIS process has
    touched the GPU), pass rank 0's JSON line through, fail if any rank fails."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SD_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


# --------------------------------------------------------------------------- the printed line (compact) and the detail record (side file)
COMPACT_LIMIT = 4096          # the driver keeps a few KB of stdout: the FINAL line must stay well below that (round 3's 24 KB line was not parsed)
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_profile", "algorithmic_flops_per_launch",
              "algorithmic_bytes_per_launch", "avg_launch_ms", "images_per_launch", "batch", "operands")


def _clip(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact_roofline(r):
    """The roofline object of the printed line: the top-level fields only (per-kernel tables stay in the detail record)."""
    if not r:
        return None
    out = {k: r[k] for k in _ROOF_KEYS if r.get(k) is not None}
    for k in ("traffic", "traffic_profile"):
        out.setdefault(k, None)
    if "kernel" in out:
        out["kernel"] = _clip(out["kernel"], 120)
    fe = r.get("front_end")
    if fe:                                    # the dominant HBM-bound kernel of the front end beside the detector's MFMA block
        out["front_end"] = {k: fe[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms")
                            if fe.get(k) is not None}
    return out


def compact_line(full, detail_path=None):
    """The ONE line bench.py prints: metric / value / config / roofline / cpu_baseline and one number per extra, < COMPACT_LIMIT bytes.
    Everything else (per-kernel tables, the extras' rooflines, the long sample texts) is in the detail record."""
    cfg = full.get("config") or {}
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": _clip(cfg.get("workload", ""), 700)}
    for k in ("lanes_per_gpu", "images_per_frame", "frames_timed", "timed_seconds", "detector_arithmetic", "sharding", "frames_per_block_per_lane",
              "frames_in_one_extraction_batch", "max_detector_boxes_in_a_frame"):
        if cfg.get(k) is not None:
            out["config"][k] = _clip(cfg[k], 160) if isinstance(cfg[k], str) else cfg[k]
    out["roofline"] = compact_roofline(full.get("roofline"))
    cpu = full.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {"value": cpu.get("value"), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                               "sample": _clip(cpu.get("sample", ""), 420)}
        if cpu.get("all_cores"):
            out["cpu_baseline"]["front_end_all_cores"] = {"value": cpu["all_cores"].get("value"), "threads": cpu["all_cores"].get("threads")}
    else:
        out["cpu_baseline"] = None
    out["extra"] = {k: (v.get("value") if isinstance(v, dict) and "value" in v else {"error": _clip(v.get("error", "?"), 80)} if isinstance(v, dict) else v)
                    for k, v in (full.get("extra") or {}).items()}
    for k in ("value_f32x3", "dtype_f32x3", "value_f32w", "dtype_f32w", "gathered_record_check", "vocabulary_broadcast_ms"):
        if full.get(k) is not None:
            out[k] = full[k]
    if detail_path:
        out["detail"] = detail_path
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:             # never print a line the driver cannot keep: drop the optional parts, longest first
        for k in ("gathered_record_check", "extra"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) < COMPACT_LIMIT:
                break
        if len(line) >= COMPACT_LIMIT:
            out["config"] = {"workload": _clip(cfg.get("workload", ""), 200)}
            if out.get("cpu_baseline"):
                out["cpu_baseline"]["sample"] = _clip(out["cpu_baseline"]["sample"], 120)
            line = json.dumps(out, separators=(",", ":"))
    return line


def write_detail(full, name="bench_detail.json"):
    """The full record next to bench.py (and under gpurun_out/ when that directory exists, so that it comes back from a GPU box)."""
    paths = [os.path.join(ROOT, name)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", name))
    written = None
    for p_ in paths:
        try:
            with open(p_, "w") as f:
                json.dump(full, f, indent=1)
            written = written or os.path.relpath(p_, ROOT)
        except OSError:
            continue
    return written


# --------------------------------------------------------------------------- synthetic inputs
def colour_from_gray(gray, rng):
    """3 unequal channels (so that cvtColor is exercised), as synth.rgbd_frame does."""
    dr = rng.integers(-3, 4, gray.shape); db = rng.integers(-3, 4, gray.shape)
    g = gray.astype(np.int16)
    return np.stack([np.clip(g + dr, 0, 255), g, np.clip(g + db, 0, 255)], axis=-1).astype(np.uint8)


def synth_timestep(synth, kind, cfg, seq, t, cut=120):
    """One frame of one synthetic sequence (a sequence stays inside its texture canvas for ~128 frames: scene cut every `cut`).
    -> dict(images (ipl, H, W, 3) u8, depth (H, W) u16 or None, boxes (k, 4) f64, mask (H, W) u8 or None, stamp)."""
    s, tt = seq + 1000 * (t // cut), t % cut
    rng = np.random.Generator(np.random.PCG64(7919 * seq + t))
    stamp = t / float(cfg["fps"])
    rows = synth.boxes_for_frame(s, tt, cfg)
    if kind == "stereo":
        left, right, _ = synth.stereo_frame_dyn(s, tt, cfg)
        return dict(images=np.stack([colour_from_gray(left, rng), colour_from_gray(right, rng)]), depth=None,
                    boxes=synth.rows_to_rects(rows), mask=None, stamp=stamp)
    rgb, depth, _ = synth.rgbd_frame_dyn(s, tt, cfg)
    return dict(images=rgb[None], depth=depth, boxes=synth.rows_to_rects(rows), mask=synth.mask_from_boxes(rows, cfg["width"], cfg["height"]), stamp=stamp)


# --------------------------------------------------------------------------- CPU baseline (oracle; rank 0, N=1 only)
def cpu_baseline(workload, cfg, kind, sensor, with_boxes, with_detector, pkg, budget_s=25.0):
    """The frame-level oracle (oracle/pipeline.py) on the host: the reference's own threading (two extraction threads per stereo
    frame, Frame.cc:87-90; everything else in the caller's thread), steady-clock per frame, 20 warm-up frames then as many
    measured frames as fit the budget (>= 200 when they do), median and mean (stereo_kitti.cc:96-170).  The detector (torch fp32
    on the host cores, the way cv::dnn runs it: DNN_TARGET_CPU) is timed on a few images and added per frame."""
    import importlib.util
    import torch
    native = os.environ.get("SD_ORACLE_NATIVE", "1") != "0"
    orc = graft.load_oracle()
    flags = "-march=x86-64-v3"
    if native:
        try:
            orc.use_native()
            flags = "-march=native"
        except Exception:
            pass
    spec = importlib.util.spec_from_file_location("sd_oracle_pipeline", os.path.join(ROOT, "oracle", "pipeline.py"))
    P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)
    synth = pkg.synth
    o = P.SequenceOracle(orc, cfg, sensor, rgb_order=True, track_last=True, threads=2 if sensor == P.SENSOR_STEREO else 1)
    warm, times = 20, []
    pool = [synth_timestep(synth, kind, cfg, 900, t) for t in range(100)]        # generation is not timed
    t_start = time.perf_counter()
    k = 0
    while True:
        fr = pool[k % len(pool)] if k < len(pool) else synth_timestep(synth, kind, cfg, 900, k)
        im0 = fr["images"][0]
        im1 = fr["images"][1] if kind == "stereo" else fr["depth"]
        t0 = time.perf_counter()
        o.track(im0, im1, fr["boxes"] if with_boxes else None, fr["stamp"])
        dt = time.perf_counter() - t0
        if k >= warm:
            times.append(dt)
        k += 1
        if len(times) >= 200 and time.perf_counter() - t_start > budget_s * 0.5:
            break
        if time.perf_counter() - t_start > budget_s and len(times) >= 30:
            break
    times.sort()
    fe_median, fe_mean = times[len(times) // 2], sum(times) / len(times)
    det_s, det_n, det_threads = 0.0, 0, 0
    if with_detector:
        yo = graft.load_yolo_oracle()
        layers, anchors = pkg.yolo.v3_layers()
        _, per = pkg.yolo.synth_weights(layers, seed=3)
        det_threads = max(1, min(torch.get_num_threads(), usable_cpus()))      # more threads than usable cores only slows torch's convolutions down
        torch.set_num_threads(det_threads)
        blob = yo.blob_from_image(pool[0]["images"][0][:, :, ::-1], 640, 480, orc.resize_linear)
        yo.torch_forward(layers, per, blob)                     # warm-up
        t0 = time.perf_counter()
        while det_n < 3 or (time.perf_counter() - t0 < 6.0 and det_n < 20):
            yo.torch_forward(layers, per, blob); det_n += 1
        det_s = (time.perf_counter() - t0) / det_n
    per_frame = fe_mean + det_s
    cores = max(2 if sensor == P.SENSOR_STEREO else 1, det_threads)
    out = {"value": round(1.0 / per_frame, 3), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "%d warm-up + %d measured consecutive frames of one synthetic sequence of the same workload through the frame-level CPU oracle "
                     "(oracle/pipeline.py over libsd_oracle, g++ -O3 %s), %d extraction thread(s) as the reference; front end median %.2f ms / mean %.2f ms"
                     % (warm, len(times), flags, 2 if sensor == P.SENSOR_STEREO else 1, fe_median * 1e3, fe_mean * 1e3) +
                     ("; detector = YOLOv3 torch-fp32 forward on %d host threads, %.0f ms / image over %d images, added per frame" % (det_threads, det_s * 1e3, det_n)
                      if with_detector else "") + "; host has %d logical cores, %d usable by this process" % (os.cpu_count() or 0, usable_cpus()),
           "front_end_ms": {"median": round(fe_median * 1e3, 3), "mean": round(fe_mean * 1e3, 3), "frames": len(times)},
           "detector_ms": round(det_s * 1e3, 2) if with_detector else None}
    return out


def pingpong_index(t, P):
    """Index of the resident time step shown at step t when P steps are resident: 0 .. P-1, P-2 .. 1, 0, 1, ... (consecutive steps always
    show neighbouring resident steps)."""
    if P <= 1:
        return 0
    m = t % (2 * P - 2)
    return m if m < P else 2 * P - 2 - m


def detector_algorithmic_bytes(layers, net_w, net_h, batch, elt=4):
    """HBM bytes a detector batch needs at least: every convolution reads its input once, writes its output once (plus the shortcut
    operand it adds in its epilogue) and reads its weights once per launch."""
    shp, act, wts = [], 0, 0
    C, h, w = 3, net_h, net_w
    for i, L in enumerate(layers):
        t = int(L["type"])
        if t == 0:                                            # yolo.CONV
            cin, hin, win = C, h, w
            st = int(L["stride"])
            h, w, C = (hin + st - 1) // st, (win + st - 1) // st, int(L["filters"])
            act += (cin * hin * win + C * h * w) * elt
            wts += C * cin * int(L["size"]) ** 2 * elt
            if i + 1 < len(layers) and int(layers[i + 1]["type"]) == 1:
                act += C * h * w * elt
        elif t == 2:                                          # route
            fr = [int(v) if int(v) >= 0 else i + int(v) for v in L["from"][:int(L["nfrom"])]]
            C, h, w = sum(shp[f][0] for f in fr), shp[fr[0]][1], shp[fr[0]][2]
        elif t == 3:                                          # upsample
            h, w = 2 * h, 2 * w
        shp.append((C, h, w))
    return act * batch + wts


def usable_cpus():
    """Host cores this process may actually run on: the affinity mask, cut by a cgroup CPU quota when there is one (the GPU box gives a
    one-GPU job a share of the host, and threads beyond it only fight each other)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.999)))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, int(q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.999)))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_all_cores(cfg, kind, sensor, with_boxes, pkg, threads, frames_per_thread=6):
    """`threads` independent camera streams at once, one per host thread (the oracle's C calls release the GIL): an all-cores
    figure for the front end only, reported beside the single-stream baseline, never instead of it."""
    import importlib.util
    import threading
    orc = graft.load_oracle()
    spec = importlib.util.spec_from_file_location("sd_oracle_pipeline", os.path.join(ROOT, "oracle", "pipeline.py"))
    P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)
    pool = [synth_timestep(pkg.synth, kind, cfg, 901, t) for t in range(frames_per_thread)]
    os_ = [P.SequenceOracle(orc, cfg, sensor, track_last=True, threads=1) for _ in range(threads)]

    def work(k):
        for fr in pool:
            os_[k].track(fr["images"][0], fr["images"][1] if kind == "stereo" else fr["depth"], fr["boxes"] if with_boxes else None, fr["stamp"])
    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    return {"value": round(threads * frames_per_thread / dt, 2), "unit": "frames/s", "threads": threads,
            "sample": "front end only: %d independent streams x %d consecutive frames, one stream per thread, %.1f s" % (threads, frames_per_thread, dt)}


# --------------------------------------------------------------------------- one workload on this rank
class Workload:
    def __init__(self, name, args, rank, world, dev, pkg, dist, vocab=None):
        import torch
        self.torch = torch
        fe, synth = pkg.frontend, pkg.synth
        self.fe, self.synth, self.pkg, self.name, self.dev, self.dist, self.rank, self.world = fe, synth, pkg, name, dev, dist, rank, world
        self.detector = name in ("stereo-yolo", "stereo-yolo-f32w", "stereo-yolo-f32x3", "stereo-yolo-f16")
        # f32 = the reference's arithmetic (cv::dnn on the CPU computes in f32); f32w = f32 with the 3 x 3 stride-1 layers as Winograd F(2x2, 3x3)
        self.det_prec = {"stereo-yolo-f16": "f16", "stereo-yolo-f32w": "f32w", "stereo-yolo-f32x3": "f32x3"}.get(name, "f32")      # f32x3: f32 operands as three bf16 limbs
        self.with_boxes = name in ("stereo-yolo", "stereo-yolo-f32w", "stereo-yolo-f32x3", "stereo-yolo-f16", "rgbd-cull", "tum-mask", "kitti-batch")
        self.kind = "stereo" if name in ("stereo-yolo", "stereo-yolo-f32w", "stereo-yolo-f32x3", "stereo-yolo-f16", "stereo", "kitti-batch") else "rgbd"
        self.bow = name == "rgbd-bow"
        self.vocab = vocab
        self.bow_history = []             # per step: the lanes' ring slots (the copies q_frame holds)
        if self.bow and vocab is None:
            raise SystemExit("rgbd-bow needs the vocabulary")
        self.cfg = synth.TUM3 if name == "tum-mask" else (synth.KITTI_STEREO if self.kind == "stereo" else synth.KITTI03_RGBD)
        self.sensor = fe.SENSOR_STEREO if self.kind == "stereo" else fe.SENSOR_RGBD
        self.ipl = 2 if self.kind == "stereo" else 1
        cfg = self.cfg
        self.W, self.H = cfg["width"], cfg["height"]
        self.strong = False                       # kitti-batch (configs[4]) is SequenceBatchWorkload
        self.S = args.lanes
        self.my_sequences = list(range(self.S))
        self.idle = False
        self.distinct = min(args.distinct, self.S)
        self.ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
        # Detector-less workloads pipeline the steps through the library's time-batched mode with a look-ahead of ONE frame: the history-free half of frame
        # t + 1 (sd_tracker_prefetch) is enqueued before frame t's call synchronises, so the GPU never waits for the host's turn-around between steps
        # (~0.1 ms of a 4.1 ms step).  Results are those of plain calls (tests/test_gpu_pipeline.py::test_time_batched_prefetch_equals_sequential).
        self.pipelined = not self.detector and os.environ.get("SD_BENCH_PIPELINE", "1") != "0"
        self.trk = fe.Tracker(self.ex, cfg, self.sensor, self.S, channels=3, rgb_order=True, track_last=True, lookahead=1 if self.pipelined else 0)
        self.batch = self.trk.batch
        self.main = torch.cuda.current_stream()
        self.prefetched = -1              # last time step whose history-free half has been enqueued
        self.prefetch_limit = None        # drain: no history-free half beyond this step is enqueued any more
        self.det = None
        if self.detector:
            # --det-split n: the frame's images go through the detector as n independent sub-batches on their own streams (one
            # sub-batch's convolutions could fill the chip while the other's drain or run their small kernels).  Same work, same
            # results, only the launch schedule differs -- and on MI355X no gain was measured, so the default stays 1.
            layers, anchors = pkg.yolo.v3_layers()
            self.n_det = max(1, min(int(args.det_split), self.S))
            while self.S % self.n_det:
                self.n_det -= 1
            self.S_det = self.S // self.n_det
            payload = pkg.yolo.synth_weights(layers, seed=3)[0]
            self.dets = []
            for _ in range(self.n_det):
                d_ = pkg.yolo.Detector(layers, anchors, 640, 480, max_batch=self.S_det, precision=self.det_prec)
                d_.load_weights(payload)
                self.dets.append(d_)
            self.det = self.dets[0]
            # (a high-priority detector stream -- torch.cuda.Stream(priority=-1) -- was measured: 1,042.4 / 1,042.9 frames/s against 1,041.5 / 1,042.5: no effect)
            self.det_streams = [torch.cuda.Stream(device=dev) for _ in range(self.n_det)]
            self.det_stream = self.det_streams[0]
            # Overlap mode (sd_yolo_set_overlap; SD_BENCH_DET_OVERLAP=1): a pass's blobFromImage and region decodes on the detector's internal streams, NMS +
            # download on a stream of their own, TWO passes kept enqueued ahead of the frame being tracked, so that the convolutions of pass t + 1 start the
            # moment pass t's last convolution ends.  MEASURED on one box, back to back (round 4, 12 and 20 timed steps): one stream / one pass ahead
            # 1,067.6 and 1,066.2 frames/s, overlap 1,060.4 and 1,057.6 -- the small kernels (3 ms per 256-image pass) cost the convolutions more when they
            # share the CUs with them than when they wait their turn (every layer is a dependent launch that ends with its slowest workgroup, and a
            # latency-bound kernel camps on a few CUs for its whole 0.4 - 0.9 ms).  Part of that loss was stream aliasing (4 hardware queues; see
            # GPU_MAX_HW_QUEUES in main()): with 16 queues, same box, 20 timed steps, off 1,068.7 / 1,067.0, on 1,070.3 / 1,068.9 -- + 0.15 %, inside the
            # noise.  So the default stays the single stream; the mode stays tested and switchable.
            self.box_streams = [torch.cuda.Stream(device=dev) for _ in range(self.n_det)]
            self.overlap = self.det_prec != "f16" and os.environ.get("SD_BENCH_DET_OVERLAP", "0") == "1"
            self.depth = 2 if self.overlap else 1
            if self.overlap:
                for d_ in self.dets:
                    d_.set_overlap(True)
            self.det_enqueued = -1            # last time step whose pass has been enqueued
            self.lookahead = True             # passes are launched ahead of the frame whose front end runs
            S_ = self.S                       # depth + 1 result slots: forward + NMS + download of frames t + 1 .. t + depth are queued before frame t's boxes are consumed
            M = fe.MAXB                       # SD_MAX_BOXES: the detector's box table and the tracker's have the same stride, nothing is cut
            slots = self.depth + 1
            self.det_dev = [dict(b=torch.zeros((S_, M, 4), dtype=torch.float64, device=dev), c=torch.zeros((S_, M), dtype=torch.int32, device=dev),
                                 f=torch.zeros((S_, M), dtype=torch.float32, device=dev), n=torch.zeros((S_,), dtype=torch.int32, device=dev)) for _ in range(slots)]
            self.det_host = [dict(b=torch.zeros((S_, M, 4), dtype=torch.float64).pin_memory(), n=torch.zeros((S_,), dtype=torch.int32).pin_memory()) for _ in range(slots)]
            self.det_ev = [[torch.cuda.Event() for _ in range(self.n_det)] for _ in range(slots)]
        self.cloud = name == "tum-mask"          # PointCloudMapping::generatePointCloud on every frame: the consumer of the semantic mask
        if self.cloud:
            self.cap_pts = ((self.W + 2) // 3) * ((self.H + 2) // 3)
            self.d_pts = torch.zeros((self.S, self.cap_pts, 16), dtype=torch.uint8, device=dev)
            self.d_cnt = torch.zeros((self.S, 2), dtype=torch.int32, device=dev)
            self.Twc64 = np.tile(np.eye(4), (self.S, 1, 1))
            self.cam = fe.make_camera(cfg)
            self.depth_factor = float(np.float32(1.0) / np.float32(cfg["depth_map_factor"]))
        self.frames = []          # per timestep: dict(images, depth, boxes, n_boxes, stamps)
        self.n_total = 0          # time steps this workload will run (prepare() adds to it)
        self.max_resident = 48     # time steps generated and parked in HBM; longer runs walk them back and forth
        self.t = 0
        self.n_boxes_seen = []
        self.max_det_boxes = 0

    def prepare(self, n_steps):
        """Generate `n_steps` more time steps and park them in HBM (distinct sequences on the host, replicated to the lanes on the device)."""
        torch, synth, cfg = self.torch, self.synth, self.cfg
        S, D = self.S, self.distinct
        reps = (S + D - 1) // D
        self.n_total += n_steps
        base = len(self.frames)
        for t in range(base, min(base + n_steps, self.max_resident)):
            per = [synth_timestep(synth, self.kind, cfg, 10 + 37 * self.rank + d, t) for d in range(D)]
            img = torch.from_numpy(np.stack([p["images"] for p in per])).to(self.dev)                  # [D, ipl, H, W, 3]
            img = img.repeat((reps, 1, 1, 1, 1))[:S].contiguous()
            dep = None
            if self.kind == "rgbd":
                dep = torch.from_numpy(np.stack([p["depth"] for p in per]).view(np.int16)).to(self.dev).repeat((reps, 1, 1))[:S].contiguous()
            bx = np.zeros((S, self.fe.MAXB, 4), np.float64); nb = np.full(S, -1, np.int32)
            if self.with_boxes and not self.detector:
                for l in range(S):
                    b = per[l % D]["boxes"]; nb[l] = len(b); bx[l, :len(b)] = b
            msk = None
            if self.cloud:
                msk = torch.from_numpy(np.stack([p["mask"] for p in per])).to(self.dev).repeat((reps, 1, 1))[:S].contiguous()
            self.frames.append(dict(images=img, depth=dep, mask=msk, boxes=bx, n_boxes=nb, stamps=np.full(S, per[0]["stamp"], np.float64)))

    def frame_at(self, t):
        """Time step t of the run.  Beyond the resident steps the sequences are walked back and forth (P-1, P-2, ..., 1, 0, 1, ...): the
        camera motion reverses, consecutive frames stay consecutive views of the same scene, and the time stamps keep increasing -- the
        work per step is that of a longer sequence, for an arbitrary --steps, within a fixed amount of HBM and of host generation time."""
        P = len(self.frames)
        if t < P:
            return self.frames[t]
        fr = dict(self.frames[pingpong_index(t, P)])
        fr["stamps"] = np.full(self.S, t / float(self.cfg["fps"]), np.float64)
        return fr

    def step(self):
        fe, torch = self.fe, self.torch
        fr = self.frame_at(self.t)
        W, H, S = self.W, self.H, self.S
        boxes, n_boxes = (fr["boxes"], fr["n_boxes"]) if self.with_boxes else (None, None)
        if self.det is not None:
            last = self.n_total - 1 if self.lookahead else self.t
            while self.det_enqueued < min(self.t + self.depth - 1, last):      # first step / after a reset: nothing was launched ahead
                self.enqueue_detector(self.det_enqueued + 1)
            k = self.t % (self.depth + 1)
            for e in self.det_ev[k]:
                e.synchronize()                   # yolo->Segmentation_(imLeft) of THIS frame is on the host (stereo_kitti.cc:107)
            nb_all = self.det_host[k]["n"].numpy()
            if (nb_all < 0).any():
                raise RuntimeError("detector post-processing on the device exceeded its capacity")
            n_boxes = nb_all.astype(np.int32)      # every box the detector kept goes on, as `SLAM.TrackStereo(imLeft, imRight, boxes, t)` does (stereo_kitti.cc:107-122);
            self.max_det_boxes = max(self.max_det_boxes, int(nb_all.max()))      # an overflow of the box tables is SD_ERR_CAPACITY, never a cut
            boxes = self.det_host[k]["b"].numpy().copy()
            if self.lookahead and self.det_enqueued < min(self.t + self.depth, last):     # one more pass goes out behind the ones in flight
                self.enqueue_detector(self.det_enqueued + 1)
        if self.pipelined:
            for tt in (self.t, self.t + 1):
                if tt > self.prefetched and tt < (self.n_total if self.prefetch_limit is None else self.prefetch_limit):
                    f2 = self.frame_at(tt)
                    self.trk.prefetch(f2["images"].data_ptr(), W * 3, W * H * 3, 1, d_depth=f2["depth"].data_ptr() if f2["depth"] is not None else 0,
                                      depth_stride=W, depth_pitch=W * H, stream=self.main.cuda_stream)
                    self.prefetched = tt
            res = self.trk.track(0, W * 3, W * H * 3, fr["stamps"], boxes=boxes, n_boxes=n_boxes, stream=self.main.cuda_stream)
        else:
            res = self.trk.track(fr["images"].data_ptr(), W * 3, W * H * 3, fr["stamps"], boxes=boxes, n_boxes=n_boxes,
                                 d_depth=fr["depth"].data_ptr() if fr["depth"] is not None else 0, depth_stride=W, depth_pitch=W * H,
                                 stream=self.main.cuda_stream)
        if self.bow:
            # Frame::ComputeBoW of mCurrentFrame and of the queued frame two steps (0.2 s at 10 fps) back -- its ring slot is still
            # q_frame's oldest entry after this step --, then SearchByBoW(that frame as the key frame, mCurrentFrame)
            cur = np.array([r.cur_slot for r in res], np.int32)
            self.bow_history.append(np.array([r.last_slot for r in res], np.int32))
            if len(self.bow_history) > 3:
                self.bow_history.pop(0)
            if len(self.bow_history) == 3:
                ref = self.bow_history[0]
                self.batch.compute_bow(self.vocab, np.concatenate([cur, ref]), 4, stream=self.main.cuda_stream)
                self.batch.search_by_bow(ref, cur, 0.7, True, stream=self.main.cuda_stream)
            else:
                self.batch.compute_bow(self.vocab, cur, 4, stream=self.main.cuda_stream)
        if self.cloud:
            self.batch.backproject_dense(np.arange(S, dtype=np.int32), fr["images"].data_ptr(), W * 3, W * H * 3, fr["depth"].data_ptr(), W, W * H,
                                         self.depth_factor, fr["mask"].data_ptr(), W, W * H, self.cam, self.Twc64, self.d_pts.data_ptr(), self.cap_pts,
                                         self.d_cnt.data_ptr(), stream=self.main.cuda_stream)
        self.t += 1
        return res

    def enqueue_detector(self, t):
        """forward (convolution stream) + postprocess_ (device NMS) + download of frame t's boxes (box stream), nothing waits on the host."""
        torch = self.torch
        W, H, S = self.W, self.H, self.S
        k = t % (self.depth + 1)
        fr = self.frame_at(t)
        d = self.det_dev[k]
        Sn = self.S_det
        for p in range(self.n_det):
            ds = self.det_streams[p].cuda_stream
            bst = self.box_streams[p] if self.overlap else self.det_streams[p]
            lo, hi = p * Sn, (p + 1) * Sn
            self.dets[p].forward_device(fr["images"].data_ptr() + lo * self.ipl * W * H * 3, W, H, W * 3, self.ipl * W * H * 3, Sn, 0.5, ds)
            self.dets[p].boxes_device(Sn, W, H, d["b"][lo:hi].data_ptr(), d["c"][lo:hi].data_ptr(), d["f"][lo:hi].data_ptr(), d["n"][lo:hi].data_ptr(), stream=bst.cuda_stream)
            with torch.cuda.stream(bst):
                self.det_host[k]["b"][lo:hi].copy_(d["b"][lo:hi], non_blocking=True)
                self.det_host[k]["n"][lo:hi].copy_(d["n"][lo:hi], non_blocking=True)
                self.det_ev[k][p].record(bst)
        self.det_enqueued = t

    def close(self):
        self.trk.close()
        if self.det is not None:
            for d_ in self.dets:
                d_.close()


def block_plan(T, S, block_frames, resident):
    """(D, [(t0, n, [resident frame index of t0 .. t0 + n - 1])]) -- D = frames per lane and block so that S * D >= block_frames, the T
    frames of a sequence walked back and forth over `resident` generated ones."""
    D = max(1, min(T, -(-block_frames // max(1, S))))
    P = min(resident, T)
    return D, [(t0, min(D, T - t0), [pingpong_index(t, P) for t in range(t0, min(t0 + D, T))]) for t0 in range(0, T, D)]


def block_schedule(T, D):
    """Frames per time block of a T-frame job whose full blocks hold about D frames: D / 8, D / 4, D / 2, D, ..., D, D / 2, D / 4, D / 8.
    Nothing overlaps the FIRST block's detector pass (the pipeline fills) or the LAST block's recurrence (it drains): with plain blocks of D that is
    a full block of each -- at 8 ranks (D = 94, three blocks) a third of the job.  The ramps make both ends an eighth of a block; every block's
    recurrence (n frames x ~1 ms) still hides behind the next block's detector pass (>= half the frames x ~0.9 ms x sequences / ranks).  A job too
    short for the ramps of D gets smaller blocks (D halved until they fit); the frames between the ramps are cut into equal blocks of at most D."""
    D = max(1, min(D, T))
    while True:
        ramp = sorted({max(1, D // 8), max(1, D // 4), max(1, D // 2)} - {D})
        if 2 * sum(ramp) + D <= T or D == 1:
            break
        D = max(1, D // 2)
    if 2 * sum(ramp) + D > T:
        ramp = []
    mid = T - 2 * sum(ramp)
    m = -(-mid // D)
    base, extra = divmod(mid, m)
    return ramp + [base + 1] * extra + [base] * (m - extra) + ramp[::-1]


def frame_shard_plan(n_seq, T, world, block_frames, resident=24):
    """BASELINE configs[4] with FRAMES -- not sequences -- sharded over the ranks (north star: "independent frames shard across the 8 GPUs").

    A frame's history-free half (detector, cvtColor, ORB extraction of both eyes, stereo matching: everything before Frame::boxTrack,
    Frame.cc:129-161, > 95 % of its time) may run on ANY rank; the recurrence (Frame.cc:162 onward, Tracking.cc:620-666, 952-959) stays with
    the rank that owns the sequence.  Time is cut into blocks of D consecutive frames of every sequence; the n_seq * n frames ("units") of a
    block, in sequence-major order u = q * n + k, are cut into `world` contiguous pieces of equal size (+- 1, the pieces rotated over the ranks
    from block to block); each rank runs the history-free half of its piece in one batch and the units' records go to the owners in ONE
    all-to-all per block.  D is chosen so that a rank's piece of a full block holds about `block_frames` units; the first and last blocks are shorter
    (block_schedule).

    -> dict(owner, lanes (per rank: the sequences of its tracker lanes, padded by repetition to S), S, D, blocks); a block = dict(t0, n, idx
    (resident frame index per k), units [(q, k)], piece [(lo, hi) per rank])."""
    if world > n_seq:
        raise ValueError("more ranks than sequences")
    owner = shard_sequences(n_seq, [T] * n_seq, world)
    seqs_of = [[q for q in range(n_seq) if owner[q] == r] for r in range(world)]
    S = max(len(x) for x in seqs_of)
    lanes = [x + [x[-1]] * (S - len(x)) for x in seqs_of]
    D = max(1, min(T, -(-block_frames * world // n_seq)))
    P = min(resident, T)
    blocks = []
    t0 = 0
    for bi, n in enumerate(block_schedule(T, D)):
        U = n_seq * n
        bounds = [(U * j) // world for j in range(world + 1)]
        piece = [(bounds[(r + bi) % world], bounds[(r + bi) % world + 1]) for r in range(world)]
        blocks.append(dict(t0=t0, n=n, idx=[pingpong_index(t, P) for t in range(t0, t0 + n)], units=[(q, k) for q in range(n_seq) for k in range(n)], piece=piece))
        t0 += n
    return dict(owner=owner, lanes=lanes, S=S, D=max(B["n"] for B in blocks), blocks=blocks, n_seq=n_seq, T=T, world=world)


def frame_shard_rank_view(plan, block, rank):
    """What `rank` does with one block of frame_shard_plan: mine = its units (q, k) in piece order; send_perm = positions of `mine` grouped by the
    owner of the unit's sequence (the all-to-all wants rows grouped by destination), send_splits[dst]; recv_units = the units it receives, source
    by source, recv_splits[src]; pool_index[k * S + lane] = row of recv_units that is frame k of the lane's sequence (a padding lane repeats one)."""
    owner, world = plan["owner"], plan["world"]
    units, piece = block["units"], block["piece"]
    lo, hi = piece[rank]
    mine = units[lo:hi]
    send_perm = sorted(range(len(mine)), key=lambda i: (owner[mine[i][0]], i))
    send_splits = [sum(1 for u in mine if owner[u[0]] == d) for d in range(world)]
    recv_units, recv_splits = [], []
    for src in range(world):
        a, b = piece[src]
        got = [u for u in units[a:b] if owner[u[0]] == rank]                                 # in the source's own order
        recv_units += got
        recv_splits.append(len(got))
    pos = {u: i for i, u in enumerate(recv_units)}
    pool_index = [pos[(q, k)] for k in range(block["n"]) for q in plan["lanes"][rank]]
    return dict(mine=mine, send_perm=send_perm, send_splits=send_splits, recv_units=recv_units, recv_splits=recv_splits, pool_index=pool_index)


def exchange_rows(dist, world, send, send_splits, recv_splits):
    """The hand-over of a block: rows of `send` (grouped by destination rank) -> the rows this rank receives, source by source.  RCCL:
    one asynchronous all_to_all_single with split sizes (every unit travels over the direct xGMI link between its worker and its owner; a block
    is tens of MB, the links idle otherwise); gloo (CPU tests, single-GPU rehearsal): the same call on host copies.  -> (rows, work or None)."""
    import torch
    n_recv = int(sum(recv_splits))
    if dist is None or (world == 1 and not os.environ.get("SD_BENCH_FORCE_DIST")):      # FORCE_DIST: the collective itself on a one-rank RCCL group (1-GPU rehearsal)
        return send, None
    out = torch.empty((n_recv,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    if dist.get_backend() == "nccl":
        return out, dist.all_to_all_single(out, send, output_split_sizes=list(recv_splits), input_split_sizes=list(send_splits), async_op=True)
    host = torch.empty(out.shape, dtype=out.dtype)
    dist.all_to_all_single(host, send.cpu().contiguous(), output_split_sizes=list(recv_splits), input_split_sizes=list(send_splits))
    out.copy_(host)
    return out, None


class SequenceBatchWorkload:
    """BASELINE configs[4] ("kitti-batch"): 11 KITTI stereo sequences x T frames with the f32 detector in the loop, FRAMES sharded over the
    ranks (frame_shard_plan).  A sequence cannot be cut into independently processed chunks -- Frame::boxTrack's ids (`max + 1`,
    Frame.cc:545-550) depend on its whole history -- but only the recurrence is bound to it.  Per block of D frames of every sequence each
    rank, as a WORKER, runs detector + sd_tracker_prefetch (cvtColor, ORB extraction of both eyes, stereo matching) on its equal share of the
    block's frames in one batch (a one-lane worker tracker: the batch entries are its consecutive "frames"), exports the results as records
    with the detector's boxes behind them, and one all-to-all moves every record to the rank that owns its sequence; as an OWNER it imports
    its sequences' records as a prefetched block (sd_tracker_import_prefetched) and runs the recurrence frame by frame (sd_tracker_track with
    no image).  One block ahead on separate streams, two blocks outstanding.  N = 1 runs the very same path (the exchange is the identity).
    Every frame's result is the sequential one (tests/test_gpu_headline.py::test_kitti_batch_as_benched*)."""

    BOX_BYTES = 2048 + 16              # behind a record: SD_MAX_BOXES x 4 f64 (cv::Rect2d) + the box count (int32, padded)

    def __init__(self, args, rank, world, dev, pkg, dist, detector=True):
        import torch
        self.torch = torch
        fe, synth = pkg.frontend, pkg.synth
        self.fe, self.synth, self.pkg, self.dev, self.rank, self.world, self.dist = fe, synth, pkg, dev, rank, world, dist
        self.name, self.kind, self.cfg, self.sensor, self.ipl = "kitti-batch", "stereo", synth.KITTI_STEREO, fe.SENSOR_STEREO, 2
        self.detector, self.det_prec, self.with_boxes, self.strong, self.bow, self.cloud = detector, "f32", True, True, False, False
        cfg = self.cfg
        self.W, self.H = cfg["width"], cfg["height"]
        n_seq, self.T = int(getattr(args, "kitti_sequences", 11)), args.kitti_frames
        self.plan = frame_shard_plan(n_seq, self.T, world, args.block_frames, 24)
        self.S, self.D = self.plan["S"], self.plan["D"]
        self.my_sequences = self.plan["lanes"][rank]
        self.n_owned = sum(1 for q in range(n_seq) if self.plan["owner"][q] == rank)
        self.views = [frame_shard_rank_view(self.plan, B, rank) for B in self.plan["blocks"]]
        self.U_max = max(len(v["mine"]) for v in self.views)
        self.total_frames_all_ranks = n_seq * self.T
        self.distinct = self.n_owned
        assert fe.MAXB * 32 + 16 == self.BOX_BYTES
        self.ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
        # owner: the lanes are this rank's sequences; worker: ONE lane whose consecutive "frames" are the batch entries of a block's piece
        self.trk = fe.Tracker(self.ex, cfg, self.sensor, self.S, channels=3, rgb_order=True, track_last=True, lookahead=self.D)
        self.worker = fe.Tracker(self.ex, cfg, self.sensor, 1, channels=3, rgb_order=True, track_last=False, lookahead=self.U_max)
        self.batch = self.trk.batch
        self.R = self.trk.record_bytes()
        assert self.R == self.worker.record_bytes() and self.R % 16 == 0
        self.RB = self.R + self.BOX_BYTES
        # The recurrence is ~20 small dependent launches per frame beside a detector that keeps every CU slot occupied: each launch waits for a slot.  On a
        # high-priority stream the slots that convolution workgroups free (one every few microseconds somewhere on the chip) go to it first.
        prio = int(os.environ.get("SD_BENCH_RECURRENCE_PRIORITY", "-1"))
        self.main = torch.cuda.Stream(device=dev, priority=prio)
        self.pre_stream = torch.cuda.Stream(device=dev)
        self.det_stream = torch.cuda.Stream(device=dev)
        self.xchg_stream = torch.cuda.Stream(device=dev)
        # (Measured with two sequences on the rank -- what a rank of an 8-GPU job owns: the recurrence then costs ~2.4 ms per frame step against the
        # detector's 1.85 and bounds the job, 820 frames/s against 1,030 with eleven.  Stream priority -1 / 0: 825 / 817.  A CU partition --
        # hipExtStreamCreateWithCUMask: K CUs for the recurrence alone, the rest for the detector -- 647 / 653 / 650 for K = 8 / 16 / 32: worse, the
        # recurrence also holds wide kernels (grid sort, projection search, the 18-array frame copy) that want the whole chip.  Enqueueing the next
        # block's pass only after a fraction f of the block's recurrence: 822 / 816 / 805 / 718 for f = 0 / 0.25 / 0.5 / 1 -- alone the recurrence still
        # takes 0.94 ms per frame step (host + ~20 launches + one synchronisation), so running it alone buys nothing.)
        self.det = None
        self.max_det_boxes = 0
        M = fe.MAXB
        if detector:
            layers, anchors = pkg.yolo.v3_layers()
            self.det = pkg.yolo.Detector(layers, anchors, 640, 480, max_batch=self.U_max, precision="f32")
            self.det.load_weights(pkg.yolo.synth_weights(layers, seed=3)[0])
            self.det_dev = dict(b=torch.zeros((self.U_max, M, 4), dtype=torch.float64, device=dev), c=torch.zeros((self.U_max, M), dtype=torch.int32, device=dev),
                                f=torch.zeros((self.U_max, M), dtype=torch.float32, device=dev), n=torch.zeros((self.U_max,), dtype=torch.int32, device=dev))
        # two blocks outstanding: records of my piece (worker side), the received rows in pool order (owner side), the boxes on the host
        self.rec = [torch.zeros((self.U_max, self.RB), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.pool = [torch.zeros((self.D * self.S, self.RB), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.host_tail = [torch.zeros((self.D * self.S, self.BOX_BYTES), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.ev_det = [torch.cuda.Event() for _ in range(2)]
        self.ev_pre = [torch.cuda.Event() for _ in range(2)]
        self.blocks = []

    def prepare(self):
        """Generate `resident` time steps of every sequence this rank touches on the host (walked back and forth up to T frames: consecutive
        frames stay consecutive views, time stamps keep increasing) and park the frames of its pieces in HBM block by block, [units, 2, H, W, 3]."""
        torch, synth, cfg = self.torch, self.synth, self.cfg
        P = min(24, self.T)
        self.P = P
        need = sorted({(q, B["idx"][k]) for B, v in zip(self.plan["blocks"], self.views) for (q, k) in v["mine"]})
        host = {}
        for q, p in need:
            fr = synth_timestep(synth, "stereo", cfg, 10 + q, p)
            host[(q, p)] = (torch.from_numpy(fr["images"]).to(self.dev), fr["boxes"])
        M = self.fe.MAXB
        for B, v in zip(self.plan["blocks"], self.views):
            imgs = torch.stack([host[(q, B["idx"][k])][0] for (q, k) in v["mine"]]).contiguous() if v["mine"] else None
            given = None
            if self.det is None:                     # --kitti-no-detector: the 3 given boxes per frame travel exactly like the detector's
                tail = np.zeros((len(v["mine"]), self.BOX_BYTES), np.uint8)
                for i, (q, k) in enumerate(v["mine"]):
                    g = np.asarray(host[(q, B["idx"][k])][1], np.float64).reshape(-1, 4)
                    tail[i, :len(g) * 32] = g.view(np.uint8).reshape(-1)
                    tail[i, M * 32:M * 32 + 4] = np.array([len(g)], np.int32).view(np.uint8)
                given = torch.from_numpy(tail).to(self.dev)
            dv = lambda a: torch.tensor(a, dtype=torch.int64, device=self.dev)
            ident = v["send_perm"] == list(range(len(v["mine"])))
            self.blocks.append(dict(t0=B["t0"], n=B["n"], idx=B["idx"], images=imgs, given=given, view=v, send_perm=None if ident else dv(v["send_perm"]),
                                    pool_index=dv(v["pool_index"])))
        del host

    def enqueue(self, bi):
        """Worker side of block bi: detector (forward + NMS on the device) and the history-free half of this rank's piece, each on its own stream;
        records + boxes packed and grouped by destination.  Nothing waits on the host."""
        torch = self.torch
        B = self.blocks[bi]
        v = B["view"]
        W, H = self.W, self.H
        U = len(v["mine"])
        k = bi & 1
        rec = self.rec[k]
        M32 = self.fe.MAXB * 32
        if U:
            if self.det is not None:
                ds = self.det_stream.cuda_stream
                d = self.det_dev
                self.det.forward_device(B["images"].data_ptr(), W, H, W * 3, 2 * W * H * 3, U, 0.5, ds)
                self.det.boxes_device(U, W, H, d["b"].data_ptr(), d["c"].data_ptr(), d["f"].data_ptr(), d["n"].data_ptr(), stream=ds)
                with torch.cuda.stream(self.det_stream):
                    rec[:U, self.R:self.R + M32].copy_(d["b"][:U].view(torch.uint8).view(U, M32), non_blocking=True)
                    rec[:U, self.R + M32:self.R + M32 + 4].copy_(d["n"][:U].view(torch.uint8).view(U, 4), non_blocking=True)
            else:
                with torch.cuda.stream(self.det_stream):
                    rec[:U, self.R:].copy_(B["given"], non_blocking=True)
            ps = self.pre_stream.cuda_stream
            self.worker.prefetch(B["images"].data_ptr(), W * 3, W * H * 3, U, stream=ps)      # one lane: U consecutive "frames"
            self.worker.export_prefetched(0, U, rec.data_ptr(), record_stride=self.RB, stream=ps)
            self.worker.discard_prefetched()
        # The exchange stream must NOT wait here: a wait queued now would hold everything queued behind it -- the hand-over of the PREVIOUS block, which is
        # consumed after this call -- until THIS block's detector pass has finished, and the detector would then idle through that block's recurrence
        # (a 13 ms hole per block in the first version's trace, 10 % of the run).  The block's events are waited for when it is consumed.
        self.ev_det[k].record(self.det_stream)
        self.ev_pre[k].record(self.pre_stream)

    def _to_pool(self, bi):
        """Owner side, on the exchange stream: the received rows in pool order (frame-major over this rank's lanes), their box tails to the host."""
        B, k = self.blocks[bi], bi & 1
        n = B["n"] * self.S
        self.torch.index_select(B["rows"], 0, B["pool_index"], out=self.pool[k][:n])
        self.host_tail[k][:n].copy_(self.pool[k][:n, self.R:], non_blocking=True)
        self.ready[k].record(self.xchg_stream)
        B["rows"] = None

    def consume_begin(self, bi):
        """-> (boxes [n * S, M, 4] f64, n_boxes [n * S]) of block bi on the host; its records are imported as the tracker's next prefetched block."""
        # The all-to-all is issued HERE, not in enqueue(): RCCL runs a communicator's collectives in issue order, and an exchange queued a block
        # early would sit -- waiting for that block's detector pass -- in front of the per-step result gathers of the block being tracked.
        torch = self.torch
        B, k = self.blocks[bi], bi & 1
        v = B["view"]
        U = len(v["mine"])
        with torch.cuda.stream(self.xchg_stream):
            self.xchg_stream.wait_event(self.ev_det[k])
            self.xchg_stream.wait_event(self.ev_pre[k])
            rec = self.rec[k]
            B["send"] = rec[:U] if B["send_perm"] is None else rec[:U].index_select(0, B["send_perm"])      # rows grouped by destination rank
            B["rows"], work = exchange_rows(self.dist, self.world, B["send"], v["send_splits"], v["recv_splits"])
            if work is not None:
                work.wait()
            self._to_pool(bi)
            B["send"] = None
            # the import runs on the exchange stream too (the library's own streams are non-blocking: nothing orders them against torch's
            # streams but an explicit stream argument); sd_tracker_track waits for the block's event
            self.trk.import_prefetched(self.pool[k].data_ptr(), B["n"], record_stride=self.RB, stream=self.xchg_stream.cuda_stream)
        self.ready[k].synchronize()
        n, M = B["n"] * self.S, self.fe.MAXB
        tail = self.host_tail[k].numpy()[:n]
        boxes = tail[:, :M * 32].copy().view(np.float64).reshape(n, M, 4)
        nb = tail[:, M * 32:M * 32 + 4].copy().view(np.int32).reshape(n)
        if (nb < 0).any():
            raise RuntimeError("detector post-processing on the device exceeded its capacity")
        if n:
            self.max_det_boxes = max(self.max_det_boxes, int(nb.max()))
        return boxes, nb

    def run(self, n_blocks=None, after_step=None, on_frame=None):
        """All (or the first n_blocks) blocks: block b + 1 is enqueued before block b's frames are tracked.  on_frame(t, lane results): test hook.
        -> the last step's lane results."""
        nb = len(self.blocks) if n_blocks is None else min(n_blocks, len(self.blocks))
        W, H, S = self.W, self.H, self.S
        res = None
        self.enqueue(0)
        for bi in range(nb):
            B = self.blocks[bi]
            if bi + 1 < nb:
                self.enqueue(bi + 1)
            bx_all, nb_all = self.consume_begin(bi)
            for k in range(B["n"]):
                t = B["t0"] + k
                res = self.trk.track(0, W * 3, W * H * 3, np.full(S, t / float(self.cfg["fps"]), np.float64), boxes=bx_all[k * S:(k + 1) * S],
                                     n_boxes=nb_all[k * S:(k + 1) * S], stream=self.main.cuda_stream)
                if on_frame is not None:
                    on_frame(t, res)
                if after_step is not None:
                    after_step()
        return res

    def close(self):
        self.trk.close()
        self.worker.close()
        if self.det is not None:
            self.det.close()


def run_sequence_batch(args, rank, world, dev, pkg, dist, detector=True):
    """kitti-batch on this rank -> (result dict on rank 0, workload object)."""
    import torch
    fe = pkg.frontend
    wl = SequenceBatchWorkload(args, rank, world, dev, pkg, dist, detector=detector)
    wl.prepare()
    batch, cap = wl.batch, wl.batch.cap
    gatherer, rec_parts, layout = make_gatherer(wl, fe, dist, world, rank, dev)
    after = (lambda: gatherer.submit(rec_parts)) if gatherer is not None else None
    wl.run(n_blocks=1, after_step=after)               # warm-up: the first block, then every lane starts over
    if gatherer is not None:
        gatherer.finish()
    wl.trk.reset()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = wl.run(after_step=after)
    if gatherer is not None:
        gatherer.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = max_over_ranks(dist, time.perf_counter() - t0, dev)
    out = None
    if rank == 0:
        frames = wl.total_frames_all_ranks
        R = res[0]
        out = {"workload": "kitti-batch", "value": round(frames / elapsed, 2), "ms_per_step": round(elapsed / wl.T * 1e3, 4), "steps": wl.T, "frames": frames,
               "timed_s": round(elapsed, 3), "lanes_per_gpu": wl.S, "images_per_frame": 2, "frames_per_block_per_lane": wl.D, "blocks": len(wl.blocks),
               "frames_in_one_extraction_batch": wl.U_max, "distinct_frames_generated_per_sequence": wl.P, "record_bytes_per_frame": wl.RB,

               "max_detector_boxes_in_a_frame": wl.max_det_boxes if wl.det is not None else None,
               "lane0_last_frame": {"N": R.N, "N_s": R.N_s, "N_d": R.N_d, "n_boxes": R.n_boxes, "track_flag": R.track_flag, "separate_ret": R.separate_ret,
                                    "n_track_matches": R.n_track_matches, "n_last_matches": R.n_last_matches}}
        if gatherer is not None:
            rec = gatherer.recv[world - 1].view(wl.S, layout["_stride"])[0].cpu().numpy()
            d = decode_record(rec, layout, cap)
            out["gathered_record_check"] = {"from_rank": world - 1, "N": d["N"], "N_s": d["N_s"], "n_boxes": d["n_boxes"],
                                            "bytes_per_frame": layout["_stride"], "finite_depths": int(np.isfinite(d["depth"]).sum())}
    wl.close()
    return out, wl


def make_gatherer(wl, fe, dist, world, rank, dev):
    """The per-step asynchronous gather of the lanes' result records to rank 0 (None without a process group)."""
    if dist is None:
        return None, None, None
    import ctypes as C
    batch = wl.batch
    cap = batch.cap
    kp_p, desc_p, cnt_p, _ = batch.results_device()
    ur_p, dep_p = C.c_void_p(), C.c_void_p(); cc = C.c_int()
    fe.check(fe.lib().sd_batch_stereo_device(batch.h, C.byref(ur_p), C.byref(dep_p), C.byref(cc)))
    fb_bytes = fe.FRAME_BOXES_BYTES
    fb_p = fe.batch_boxes_device(batch)
    layout = record_layout(cap, fb_bytes)
    stride = layout["_stride"]
    S, ipl = wl.S, wl.ipl
    gatherer = ResultGather(dist, world, rank, S * stride, dev)
    stage2d = gatherer.stage.view(S, stride)
    srcs = {"count": (cnt_p, 4), "fb": (fb_p, fb_bytes), "kp": (kp_p, cap * 28), "desc": (desc_p, cap * 32), "uright": (ur_p.value, cap * 4), "depth": (dep_p.value, cap * 4)}
    rec_parts = []
    for k in RECORD_FIELDS:
        ptr, nbytes = srcs[k]
        src = fe.as_torch_u8(ptr, S * ipl * nbytes).view(S * ipl, nbytes)[::ipl]        # the lanes' current (left) slots
        o, n = layout[k]
        rec_parts.append((stage2d[:, o:o + n], src))
    return gatherer, rec_parts, layout


def run_workload(name, args, rank, world, dev, pkg, dist, headline, vocab=None):
    """-> dict with the measured figures of one workload (rank 0 fills everything, other ranks only take part)."""
    import torch
    fe = pkg.frontend
    if name == "kitti-batch":
        return run_sequence_batch(args, rank, world, dev, pkg, dist, detector=not args.kitti_no_detector)
    wl = Workload(name, args, rank, world, dev, pkg, dist, vocab=vocab)
    # extras: the detector-less workloads step in 2 - 5 ms, so 12 steps would be a 30 ms sample (one allocator hiccup moved `rgbd` by a quarter);
    # they run 5 x as many steps behind 3 warm-up steps
    steps, warm = (args.steps, args.warmup) if headline else ((args.extra_steps, 1) if name.startswith("stereo-yolo") else (5 * args.extra_steps, 3))
    prof_steps = 4 if (not args.no_profile and (headline or name in ("stereo", "rgbd-bow", "stereo-yolo-f32w", "stereo-yolo-f32x3"))) else 0
    # every timed step enqueues ONE detector pass `depth` frames ahead: the run needs that many frames beyond the last timed / profiled step, or the
    # last timed steps would find nothing left to enqueue and the timed region would hold fewer passes than steps
    wl.prepare(1 + warm + steps + prof_steps + max(1, getattr(wl, "depth", 1)) + (2 if getattr(wl, "pipelined", False) else 0))
    batch = wl.batch
    cap = batch.cap
    gatherer, rec_parts, layout = make_gatherer(wl, fe, dist, world, rank, dev)

    def one_step():
        r = wl.step()
        if gatherer is not None:
            gatherer.submit(rec_parts)
        return r

    one_step()                               # priming: frame 0 of every lane (initialisation, untimed)
    for _ in range(warm):
        one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = one_step()
    if gatherer is not None:
        gatherer.finish()                    # the last step's gather is inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = max_over_ranks(dist, time.perf_counter() - t0, dev)

    out = None
    if rank == 0:
        frames = world * wl.S * steps
        value = frames / elapsed
        R = res[0]
        out = {"workload": name, "value": round(value, 2), "ms_per_step": round(elapsed / steps * 1e3, 4), "steps": steps, "frames": frames,
               "timed_s": round(elapsed, 3), "lanes_per_gpu": wl.S, "images_per_frame": wl.ipl,
               "max_detector_boxes_in_a_frame": wl.max_det_boxes if wl.det is not None else None,
               "lane0_last_frame": {"N": R.N, "N_s": R.N_s, "N_d": R.N_d, "n_boxes": R.n_boxes, "track_flag": R.track_flag, "separate_ret": R.separate_ret,
                                    "n_track_matches": R.n_track_matches, "n_last_matches": R.n_last_matches}}
        if gatherer is not None:               # rank 0 decodes a record it received from the LAST rank: the gather carries usable data
            rec = gatherer.recv[world - 1].view(wl.S, layout["_stride"])[0].cpu().numpy()
            d = decode_record(rec, layout, cap)
            out["gathered_record_check"] = {"from_rank": world - 1, "N": d["N"], "N_s": d["N_s"], "n_boxes": d["n_boxes"],
                                            "bytes_per_frame": layout["_stride"], "finite_depths": int(np.isfinite(d["depth"]).sum())}
    # ---- separate, untimed pass: per-kernel durations (hipEvents on the kernels' own stream) for the roofline
    if prof_steps:
        if getattr(wl, "pipelined", False):      # the per-kernel pass times plain steps (the library's kernel timers belong to the tracker's own workspace)
            wl.prefetch_limit = wl.prefetched + 1
            while wl.t <= wl.prefetched:
                wl.step()
            wl.pipelined = False
        batch.set_profiling(True); batch.reset_kernel_times()
        det_ms = None
        if wl.det is not None:                 # the detector alone, nothing else on the GPU: the MFMA block
            torch.cuda.synchronize()
            fr = wl.frame_at(wl.t)
            Sn = wl.S_det

            def det_pass(p):
                wl.dets[p].forward_device(fr["images"].data_ptr() + p * Sn * wl.ipl * wl.W * wl.H * 3, wl.W, wl.H, wl.W * 3, wl.ipl * wl.W * wl.H * 3, Sn, 0.5,
                                          wl.det_streams[p].cuda_stream)
            for p in range(wl.n_det):
                det_pass(p)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = [torch.cuda.Event(enable_timing=True) for _ in range(wl.n_det)]
            e0.record(wl.det_streams[0])
            for p in range(1, wl.n_det):
                wl.det_streams[p].wait_event(e0)                       # every sub-batch starts behind the same mark
            for _ in range(3):
                for p in range(wl.n_det):
                    det_pass(p)
            for p in range(wl.n_det):
                e1[p].record(wl.det_streams[p])
            torch.cuda.synchronize()
            det_ms = max(e0.elapsed_time(e) for e in e1) / 3
            wl.det_enqueued = wl.t - 1          # the profiled steps below run exactly like the timed ones (the passes ahead are enqueued again)
        for _ in range(prof_steps):
            wl.step()
        torch.cuda.synchronize()
        batch.sync()
        kt = batch.kernel_times()
        batch.set_profiling(False)
        if rank == 0:
            cfg = wl.cfg
            alg = algorithmic_bytes(wl.W, wl.H, wl.ex.mvInvScaleFactor, cfg["n_features"], channels=3)
            n_img = wl.S * wl.ipl
            hbm = {k: v for k, v in kt.items() if v[1] > 0 and alg.get(k, 0) > 0}
            dom = max(hbm, key=lambda k: hbm[k][0])
            ms, launches = hbm[dom]
            per_step_launches = launches / prof_steps
            per_launch = alg[dom] * n_img / per_step_launches
            avg_ms = ms / launches
            achieved = per_launch / (avg_ms * 1e-3) / 1e9
            pmc = load_pmc_traffic(name)
            table = kernel_roofline_table(kt, prof_steps, alg, n_img, wl.S, pmc)
            e = pmc.get(dom) or pmc.get(KERNEL_SYMBOL.get(dom, dom))
            traffic = e.get("hbm_bytes_per_launch") if (e and e.get("batch_images") == n_img) else None
            traffic_profile = e.get("profile") if traffic is not None else None
            fe_ms = sum(v[0] for v in kt.values()) / prof_steps
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_profile": traffic_profile, "algorithmic_bytes_per_launch": int(per_launch),
                    "avg_launch_ms": round(avg_ms, 4), "images_per_launch": int(n_img / per_step_launches),
                    "kernels": table,
                    "kernels_ms_per_step": {k: round(v[0] / prof_steps, 4) for k, v in kt.items() if v[1] > 0},
                    "front_end_kernels_ms_per_step": round(fe_ms, 4),
                    "front_end_algorithmic_GBs_while_running": round(alg["image_total"] * n_img / (fe_ms * 1e-3) / 1e9, 2),
                    "pipeline_achieved_GBs": round(alg["image_total"] * wl.ipl * out["value"] / world / 1e9, 2),
                    "measured": "separate untimed pass of %d steps run like the timed ones, hipEvents around every kernel on its own stream" % prof_steps,
                    "note": "the step is bound by the detector (the MFMA object above this one): the front-end kernels run beside it on their own stream, "
                            "their times here are CONTENDED (CU slots held by convolution workgroups) -- the uncontended per-kernel figures are under extra.stereo.roofline"
                            if wl.det is not None else None}
            if det_ms is not None:
                # the roofline counts the MFMA FLOPs the mode EXECUTES: in f32w the Winograd layers run 16 multiplies per 2 x 2 block and
                # (filter, channel) pair instead of 36, and the direct-convolution count is reported beside it as "nominal"
                nominal = wl.det.flops()
                fl = wl.det.mfma_flops()
                prec = wl.det_prec
                tf = fl * wl.S / (det_ms * 1e-3) / 1e12
                frac = tf / MFMA_PEAK_TFLOPS[prec]
                limb = None
                if prec == "f32x3":
                    # two matrix-pipe rates in one pass: the limb layers execute bf16 MFMAs (six per f32 product block), the first four layers f32 MFMAs.
                    # `achieved` is the bf16 rate over the WHOLE pass time, `frac` the share of that time the matrix pipe must be busy at its peaks
                    fb = wl.det.mfma_flops_bf16()
                    limb = {"bf16_gflop_per_image": round(fb / 1e9, 2), "f32_gflop_per_image": round(fl / 1e9, 2),
                            "f32_mfma_tflops": round(tf, 2)}
                    frac = (fb / (MFMA_PEAK_TFLOPS["f32x3"] * 1e12) + fl / (MFMA_PEAK_TFLOPS["f32"] * 1e12)) * wl.S / (det_ms * 1e-3)
                    tf = fb * wl.S / (det_ms * 1e-3) / 1e12
                    limb["executed_bf16_flops_per_launch"] = int(fb * wl.S)
                # The kernel that decides this workload is the detector's convolution (k_conv_f32 / the f16 conv kernels: > 90 % of the GPU
                # time), so IT is the roofline object; one "launch" = the convolution launches of one detector batch, timed alone on the
                # detector's stream.  The dominant HBM-bound kernel of the front end stays beside it under "front_end".
                n_conv = int((wl.det.layers["type"] == 0).sum())                  # yolo.CONV
                det_traffic, det_prof = None, None
                e = pmc.get("k_conv_f32" if prec in ("f32", "f32w") else "")
                if e and e.get("batch_images") == n_img and prec == "f32":
                    det_traffic = int(e["hbm_bytes_per_launch"]) * n_conv * wl.n_det       # the profile's launches are sub-batch launches
                    det_prof = e.get("profile")
                if e and e.get("batch_images") == n_img and prec == "f32w" and "k_wino_input" in pmc and "k_wino_gemm_f32" in pmc:
                    n_w = wl.det.winograd_layers()
                    det_traffic = (int(e["hbm_bytes_per_launch"]) * (n_conv - n_w) + (int(pmc["k_wino_input"]["hbm_bytes_per_launch"])
                                   + int(pmc["k_wino_gemm_f32"]["hbm_bytes_per_launch"])) * n_w) * wl.n_det
                    det_prof = e.get("profile")
                top = {"bound": "mfma", "kernel": "k_conv_f32 x %d launches = the convolutions of one %d-image detector batch" % (n_conv * wl.n_det, wl.S) if prec == "f32"
                                                  else ("k_conv3x3_b3 / k_conv_b3 (bf16 limbs) + k_conv_f32 (first layers): the %d convolutions of one %d-image detector batch" % (n_conv * wl.n_det, wl.S) if prec == "f32x3"
                                                        else "k_conv_f32 / k_wino_input + k_wino_gemm_f32: the %d convolutions of one %d-image detector batch" % (n_conv * wl.n_det, wl.S) if prec == "f32w"
                                                        else "the f16 convolution kernels of one %d-image detector batch" % wl.S),
                       "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS[prec], "unit": "TFLOP/s", "frac": round(frac, 4), "limb_mode": limb,
                       "traffic": det_traffic, "traffic_profile": det_prof, "algorithmic_flops_per_launch": int(nominal * wl.S), "avg_launch_ms": round(det_ms, 3),
                       "executed_mfma_flops_per_launch": int(fl * wl.S),        # f32-MFMA FLOPs the mode runs (f32w: fewer than nominal; f32x3: the first layers only, the limb FLOPs are under limb_mode)
                       "algorithmic_bytes_per_launch": detector_algorithmic_bytes(wl.det.layers, wl.det.net_w, wl.det.net_h, wl.S, 2 if prec == "f16" else 4),
                       "operands": prec, "images_per_s": round(wl.S / (det_ms * 1e-3), 1), "gflop_per_image": round(nominal / 1e9, 2), "batch": wl.S,
                       "nominal_direct_convolution": {"gflop_per_image": round(nominal / 1e9, 2), "tflops": round(nominal * wl.S / (det_ms * 1e-3) / 1e12, 1)},
                       "measured": "3 detector passes alone (%d sub-batch(es) of %d images on their own streams, as in the timed steps) between events on those streams (untimed pass)" % (wl.n_det, wl.S_det),
                       "weights": "synthetic (yolov3.weights is a download that never was in the reference)",
                       "front_end": roof}
                roof = top
            out["roofline"] = roof
    wl.close()
    return out, wl


WORKLOAD_TEXT = {
    "stereo-yolo": "KITTI stereo 1241x376 colour pairs, 2000 feat/image: YOLOv3 (640x480, f32 as the reference's cv::dnn, synthetic weights) on the left image -> "
                   "boxes -> TrackStereo = cvtColor + 2x ORB extract + stereo match + boxTrack + firstSeparate + TrackHomo (SearchByProjection vs the queued frame "
                   "> 0.2 s back, H/F fit) + Separate + UpdateFrame + SearchByProjection vs the last frame (BASELINE configs[2])",
    "stereo-yolo-f32w": "the same chain with the detector's 3 x 3 stride-1 layers (>= 64 input channels) computed as Winograd F(2x2, 3x3), still f32 operands and f32 "
                        "accumulation: 2.25 x fewer MFMA FLOPs on those layers, same layer tolerance (2e-5 relative L2) and same 32 / 32 box-set equality against the "
                        "torch-fp32 oracle as the f32 mode (tests/test_gpu_yolo.py); kept beside the headline, whose detector computes the direct sums",
    "stereo-yolo-f32x3": "the same chain with the detector's >= 64-filter layers computed on three bf16 limbs per f32 operand (x = hi + mid + lo exactly, six exact limb "
                         "products per product, f32 accumulation on v_mfma_f32_32x32x16_bf16): what is dropped is <= 2^-23 of a product; same layer tolerance and same 32 / 32 "
                         "box-set equality as the f32 mode (tests/test_gpu_yolo.py); kept beside the headline",
    "stereo-yolo-f16": "the same chain with the detector in its throughput mode (f16 operands, f32 accumulation): its box sets differ from the f32 reference's "
                       "(tests/test_gpu_yolo.py counts them), so this is NOT the parity configuration",
    "stereo": "KITTI stereo 1241x376 colour pairs, 2000 feat/image: cvtColor + 2x ORB extract + stereo match + projection match vs the last frame, no detector / boxes",
    "rgbd": "KITTI-03 RGB-D 1241x376, 2000 feat/frame: cvtColor + ORB extract + RGB-D stereo + projection match vs the last frame, no semantic mask (BASELINE configs[1])",
    "rgbd-bow": "KITTI-03 RGB-D 1241x376, 2000 feat/frame: the rgbd chain + ComputeBoW (6-level x 10-way descent of the RCCL-broadcast vocabulary, BowVector / "
                "FeatureVector) of the current frame and of the queued frame 0.2 s back + SearchByBoW between them",
    "rgbd-cull": "KITTI-03 RGB-D 1241x376, 2000 feat/frame, 3 given boxes per frame: extract + match + boxTrack + firstSeparate + TrackHomo + Separate + UpdateFrame",
    "tum-mask": "TUM3 RGB-D 640x480, 1000 feat/frame, DepthMapFactor 5000, 30 fps, mask + 3 boxes per frame: extract + match + cull + dense back-projection of the "
                "pixels outside (dynamic box AND mask) (BASELINE configs[3])",
    "kitti-batch": "11 synthetic KITTI stereo colour sequences x %d frames (BASELINE configs[4]), the configs[2] chain on every frame (YOLOv3 f32 -> boxes -> "
                   "TrackStereo with the cull); FRAMES sharded over the ranks: detector + extraction + stereo matching of a time block's frames dealt evenly to "
                   "all ranks (sd_tracker_prefetch -> records), one all-to-all per block to the sequence owners (sd_tracker_import_prefetched), the per-stream "
                   "recurrence frame by frame on the owner; results identical to the sequential run",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lanes", "--batch", type=int, default=256, dest="lanes", help="independent camera streams per GPU = frames per step per GPU")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic sequences generated on the host (replicated over the lanes on the device)")
    ap.add_argument("--workload", choices=WORKLOADS, default="stereo-yolo")
    ap.add_argument("--extra", default="auto", help="comma-separated workloads also run (short) and reported under 'extra'; 'auto' = " + ",".join(AUTO_EXTRAS) + " at N=1, none otherwise; 'none'")
    ap.add_argument("--extra-steps", type=int, default=12)
    ap.add_argument("--kitti-frames", type=int, default=256)
    ap.add_argument("--kitti-sequences", type=int, default=11, help="kitti-batch: sequences of the job (BASELINE configs[4]: KITTI 00-10 = 11)")
    ap.add_argument("--block-frames", type=int, default=128, help="kitti-batch: frames per rank whose detector pass / extraction / stereo matching form one batch "
                    "(a time block = ceil(block_frames * ranks / sequences) consecutive frames of every sequence, its frames dealt evenly to the ranks)")
    ap.add_argument("--kitti-no-detector", action="store_true", help="kitti-batch with the 3 given boxes per frame instead of the detector")
    ap.add_argument("--det-split", type=int, default=1, help="sub-batches the detector processes a step's images in, each on its own stream "
                    "(measured on MI355X: 1 -> 995.5, 2 -> 995.1, 4 -> 989.7 frames/s: the convolutions' drain phases are not worth filling)")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of host time for the CPU-baseline sample (0 = skip)")
    ap.add_argument("--detail", default="bench_detail.json", help="file (next to bench.py, and under gpurun_out/ when present) that receives the FULL record; "
                    "the printed line is its compact form")
    ap.add_argument("--no-profile", action="store_true", help="skip the separate per-kernel pass (no roofline block)")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:                    # start the ranks ourselves, before anything touches the GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, world))

    # HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and two streams on one queue serialise.  A process that has
    # run several workloads has created dozens of streams (every sd_batch, detector and torch.cuda.Stream owns one), and which of the NEXT workload's
    # streams then share a queue is luck: kitti-batch after the headline ran 8 % slower than alone (957 against 1,042 frames/s on one box) because its detector
    # stream and its extraction / recurrence streams had landed on one queue; with 8 queues 998, with 16 1,049.  Set before the runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    if os.environ.get("SD_BENCH_SINGLE_DEVICE"):     # rehearsal only: several ranks share cuda:0 (gloo backend)
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d but only %d HIP devices are visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rccl_ranks = 1
    if world > 1 or os.environ.get("SD_BENCH_FORCE_DIST"):     # FORCE_DIST: one-rank RCCL group, rehearses the collective calls on a 1-GPU box
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("SD_BENCH_BACKEND", "nccl")            # "gloo" only for single-GPU rehearsal
        if backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist_mod.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_mod
        rccl_ranks = dist.get_world_size()
        if rccl_ranks != args.gpus:
            raise SystemExit("only %d of %d ranks joined" % (rccl_ranks, args.gpus))

    pkg = graft.load_package()
    fe, synth = pkg.frontend, pkg.synth
    if fe.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")

    # ---- start-up collective: the packed ORB vocabulary (synthetic k=10, L=6 tree in ORBvoc.txt's shape; the real file
    # is a download that never was in the reference), rank 0 -> all over RCCL; every rank adopts the received buffer.
    voc_ms = None
    if rank == 0:
        voc0 = fe.Vocabulary.from_nodes(synth.vocabulary(k=10, L=6, seed=1234))
        vptr, voc_bytes = voc0.packed_device()
        n_nodes_t = torch.tensor([voc0.info()["n_nodes"]], dtype=torch.int64, device=dev)
    else:
        n_nodes_t = torch.zeros(1, dtype=torch.int64, device=dev)
    if dist is not None:
        dist.broadcast(n_nodes_t, src=0)
        voc_bytes = fe.Vocabulary.packed_bytes(int(n_nodes_t.item()))
        voc_buf = fe.as_torch_u8(vptr, voc_bytes) if rank == 0 else torch.empty((voc_bytes,), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(voc_buf, src=0)
        torch.cuda.synchronize()
        voc_ms = (time.perf_counter() - t0) * 1e3
        vocab = voc0 if rank == 0 else fe.Vocabulary.from_packed_device(voc_buf.data_ptr(), voc_bytes)
    else:
        vocab = voc0
    assert vocab.info()["n_nodes"] == int(n_nodes_t.item()) and vocab.info()["k"] == 10 and vocab.info()["L"] == 6

    head, wl = run_workload(args.workload, args, rank, world, dev, pkg, dist, headline=True, vocab=vocab)
    extras = {}
    names = []
    if args.extra == "auto":
        # the other BASELINE configs (configs[1] rgbd, configs[3] tum-mask, configs[4] kitti-batch), the detector-less stereo front end (the HBM-side
        # kernel table) and the two other f32-class detector chains (value_f32x3, value_f32w); every other workload only on request (--extra a,b,c)
        names = [w for w in AUTO_EXTRAS if w != args.workload] if world == 1 else []
    elif args.extra != "none":
        names = [w for w in args.extra.split(",") if w]
    for w in names:
        if w not in WORKLOADS:
            raise SystemExit("unknown workload " + w)
        try:
            o, _ = run_workload(w, args, rank, world, dev, pkg, dist, headline=False, vocab=vocab)
        except Exception as exc:                     # an extra must never cost the headline its line
            if world > 1:
                raise                                 # ... except where ranks would fall out of step
            extras[w] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            continue
        if rank == 0:
            extras[w] = {"value": o["value"], "unit": "frames/s", "ms_per_step": o["ms_per_step"], "steps": o["steps"], "lanes_per_gpu": o["lanes_per_gpu"],
                         "scaling": "strong" if w == "kitti-batch" else "weak",
                         "workload": WORKLOAD_TEXT[w] % args.kitti_frames if w == "kitti-batch" else WORKLOAD_TEXT[w], "lane0_last_frame": o["lane0_last_frame"]}
            for k in ("roofline", "frames", "frames_per_block_per_lane", "frames_in_one_extraction_batch", "blocks", "distinct_frames_generated_per_sequence",
                      "max_detector_boxes_in_a_frame", "record_bytes_per_frame"):
                if k in o:
                    extras[w][k] = o[k]

    if rank == 0:
        cpu = None
        if world == 1 and args.cpu_budget > 0:
            P_SENSOR = {"stereo": 1, "rgbd": 2}[wl.kind]
            cpu = cpu_baseline(args.workload, wl.cfg, wl.kind, P_SENSOR, wl.with_boxes or wl.detector, wl.detector, pkg, budget_s=args.cpu_budget)
            nthr = max(1, min(usable_cpus(), 64))
            cpu["all_cores"] = cpu_all_cores(wl.cfg, wl.kind, P_SENSOR, wl.with_boxes or wl.detector, pkg, nthr)
        text = WORKLOAD_TEXT[args.workload] % args.kitti_frames if args.workload == "kitti-batch" else WORKLOAD_TEXT[args.workload]
        if wl.with_boxes:            # the two choices of this build that the reference leaves undefined / outside the boundary
            text += "; spec Q9: the stereo constructor's box split is the RGB-D constructor's (the reference's call is commented out, Frame.cc:166); spec Q14: no SLAM back end -- " \
                    "identity pose prior, map points = the frame's own stereo points, mState == OK from a stream's third frame"
        cull = wl.with_boxes
        out = {
            "metric": "tracking frames/sec (extract+match%s), %s %dx%d" % ("+dynamic-cull" if cull else "", "KITTI" if wl.cfg is not synth.TUM3 else "TUM3", wl.W, wl.H),
            "value": head["value"], "unit": "frames/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": head["steps"], "warmup": args.warmup if not wl.strong else 0,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong" if wl.strong else "weak",
            "vs_baseline": None, "dtype": "u8" + ("+" + wl.det_prec if wl.detector else ""), "data": "synthetic",
            "config": {"workload": text, "lanes_per_gpu": head["lanes_per_gpu"], "frames_per_step_per_gpu": head["lanes_per_gpu"], "images_per_frame": wl.ipl,
                       "distinct_sequences_per_gpu": wl.distinct, "timed_seconds": head["timed_s"], "frames_timed": head["frames"],
                       "detector_arithmetic": ("%s operands, f32 accumulation (v_mfma_f32_32x32x%s)" % (wl.det_prec[:3], "16_f16" if wl.det_prec == "f16" else ("16_bf16 on three bf16 limbs per operand" if wl.det_prec == "f32x3" else "2_f32"))
                                               + ("; 3 x 3 stride-1 layers as Winograd F(2x2, 3x3)" if wl.det_prec == "f32w" else "")) if wl.detector else None,
                       "lane0_last_frame": head["lane0_last_frame"],
                       "sharding": ("single GPU" if world == 1 else
                                    "frames of a time block dealt evenly to all ranks (history-free half), one RCCL all-to-all per block to the sequence owners "
                                    "(recurrence); per-step async gather of the result records to rank 0" if wl.strong else
                                    "independent lanes per rank, no data-path collective; per-step async gather of the result records to rank 0")},
            "roofline": head.get("roofline"), "cpu_baseline": cpu, "extra": extras,
        }
        for k in ("max_detector_boxes_in_a_frame", "frames_per_block_per_lane", "frames_in_one_extraction_batch", "blocks",
                  "distinct_frames_generated_per_sequence"):
            if k in head:
                out["config"][k] = head[k]
        if "gathered_record_check" in head:
            out["gathered_record_check"] = head["gathered_record_check"]
        out["vocabulary"] = {"nodes": vocab.info()["n_nodes"], "words": vocab.info()["n_words"], "packed_bytes": int(voc_bytes)}
        if voc_ms is not None:
            out["vocabulary_broadcast_ms"] = round(voc_ms, 3)
        x3 = extras.get("stereo-yolo-f32x3")
        if args.workload == "stereo-yolo" and x3 and "value" in x3:
            # the same chain with the detector's f32 operands carried as three bf16 limbs (tests/test_gpu_yolo.py holds it to the direct mode's bars and
            # to a float64 forward): a second figure beside `value`, which stays the direct f32 sums
            out["value_f32x3"] = x3["value"]
            out["dtype_f32x3"] = "u8+f32 operands as 3xbf16 limbs, f32 accumulate"
        xw = extras.get("stereo-yolo-f32w")
        if args.workload == "stereo-yolo" and xw and "value" in xw:
            # ... and with the 3 x 3 stride-1 layers as Winograd F(2x2, 3x3) in f32: the mode CLOSEST to a float64 forward of the three (same test)
            out["value_f32w"] = xw["value"]
            out["dtype_f32w"] = "u8+f32, Winograd F(2x2,3x3) on the 3x3 stride-1 layers"
        detail = write_detail(out, args.detail)
        print(compact_line(out, detail))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
