#!/usr/bin/env python3
"""bench.py — tracking front-end frames/s (extract + match [+ dynamic cull]) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic frames that are already
resident in HBM.  At N=1 the workload is BASELINE.json configs[1]:
    KITTI-03 RGB-D 1241x376, 2000 features/frame, ORB extract + match, no semantic mask
i.e. per frame (reference call stack SURVEY 3.2, 3-argument TrackRGBD):
    cvtColor RGB->gray                      Tracking.cc:256-269
    ORBextractor::operator()                ORBextractor.cc:1043-1105
    depth scaling + ComputeStereoFromRGBD   Tracking.cc:271-272, Frame.cc:1051-1072
    AssignFeaturesToGrid                    Frame.cc:463-478
    UnprojectStereo of the frame's points   Frame.cc:1074-1088
    SearchByProjection(cur, last, th=15)    ORBmatcher.cc:1485-1627 (th: Tracking.cc:990-994)
    mLastFrame = Frame(mCurrentFrame)       (slot copy)
`--workload stereo` runs BASELINE configs[2] without the detector: 2x extract + ComputeStereoMatches
+ the same grid/unproject/projection match (th=7).

Multi-GPU (driver launches one rank per GPU through torch.distributed.run): independent frames are
sharded, every rank runs the same per-GPU batch (weak scaling), there is no data-path collective.
Start-up: RCCL broadcast of the packed ORB vocabulary (synthetic k=10, L=6 tree, ~60 MB: sd_vocab) from rank 0; every rank adopts it.
Per step: asynchronous gather (to rank 0) of the fixed-stride per-frame result records, overlapped with the next step.  value = frames of ALL ranks / max
rank time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming)


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


def algorithmic_bytes(width, height, inv_scale, n_features):
    """Compulsory HBM bytes per image and per kernel: each plane read/written once."""
    sizes = level_sizes(width, height, inv_scale)
    interior = [w * h for (w, h) in sizes]
    padded = [(w + 38) * (h + 38) for (w, h) in sizes]
    b = {
        "k_pyr_level0": width * height + padded[0],
        "k_pyr_level": sum(interior[:-1]) + sum(padded[1:]),
        "k_fast_cells": sum(interior),
        "k_blur": 2 * sum(interior),
        "k_orient": n_features * 749,
        "k_describe": n_features * 512 + n_features * 32,
        "k_quadtree": 0,
    }
    b["image_total"] = (width * height + sum(padded) + sum(interior[:-1]) + 3 * sum(interior) + n_features * 749 +
                        n_features * 512 + n_features * 60)
    return b


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


class ResultGather:
    """Per-rank result gather to rank 0 (north star: "per-rank result gather"), overlapped with the next step.

    The records (keypoints + descriptors, fixed stride) are first copied into a staging tensor on the compute
    stream, then `dist.gather(..., dst=0, async_op=True)` moves them while the next step computes.  A gather to one
    root uses the 7 direct xGMI links into rank 0 in parallel; an all_gather would move 7x the bytes into every
    rank (57 GB/s per GPU at 59k frames/s) for no consumer."""

    def __init__(self, dist, world, rank, nbytes, device):
        import torch
        self.dist, self.world, self.rank = dist, world, rank
        self.stage = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        self.recv = [torch.empty((nbytes,), dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
        self.work = None

    def submit(self, parts):
        if self.work is not None:
            self.work.wait()                     # previous gather must have drained the staging tensor
        off = 0
        for p in parts:
            n = p.numel()
            self.stage[off:off + n].copy_(p, non_blocking=True)
            off += n
        self.work = self.dist.gather(self.stage, gather_list=self.recv, dst=0, async_op=True)

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None


# --------------------------------------------------------------------------- CPU baseline (oracle; rank 0, N=1 only)
def cpu_baseline(workload, cfg, n_frames, seq):
    orc = graft.load_oracle()
    pkg = graft.load_package()
    synth, fe = pkg.synth, pkg.frontend
    cam10 = fe.camera_array(fe.make_camera(cfg))
    I = np.eye(4, dtype=np.float32)
    distinct = min(n_frames, 64)                 # generating a frame costs more than extracting it: cycle 64 distinct ones
    pool = [synth.rgbd_frame(seq, t, cfg) if workload == "rgbd" else synth.stereo_frame(seq, t, cfg) for t in range(distinct)]
    frames = [pool[t % distinct] for t in range(n_frames)]
    th = 15.0 if workload == "rgbd" else 7.0
    exL = orc.Extractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    exR = orc.Extractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    return _cpu_walk(orc, workload, cfg, frames, exL, exR, cam10, I, th)


def _cpu_walk(orc, workload, cfg, frames, exL, exR, cam10, I, th):
    last = None
    n_frames = len(frames)
    t0 = time.perf_counter()
    for fr in frames:
        if workload == "rgbd":
            rgb, depth, _ = fr
            gray = orc.cvt_gray(rgb, 1)
            kp, desc = exL(gray)
            dep32 = orc.depth_to_f32(depth, float(np.float32(1.0) / np.float32(cfg["depth_map_factor"])))
            ur, dep = orc.stereo_from_rgbd(kp, dep32, cfg["bf"])
        else:
            l, r, _ = fr
            kp, desc = exL(l)
            kpR, descR = exR(r)
            ur, dep, _, _ = orc.stereo_matches(exL, exR, kp, desc, kpR, descR, cfg["bf"], cfg["fx"])
        xw, valid = orc.unproject(kp, dep, cam10, I)
        if last is not None:
            orc.search_by_projection(kp, desc, ur, last[0], last[1], last[2], last[3], I, I, cam10, exL.scale, th)
        else:
            orc.grid_cells(kp, cam10)
        last = (kp, desc, xw, valid)
    dt = time.perf_counter() - t0
    return n_frames / dt, dt


def cpu_baseline_all_cores(workload, cfg, seq, threads, frames_per_thread=48):
    """The same walk on `threads` host threads at once, one independent frame stream per thread ("an all-cores run,
    one frame per core"); the oracle's C calls release the GIL.  Reported beside the single-stream figure, never instead of it."""
    import threading
    orc = graft.load_oracle()
    pkg = graft.load_package()
    synth, fe = pkg.synth, pkg.frontend
    cam10 = fe.camera_array(fe.make_camera(cfg))
    I = np.eye(4, dtype=np.float32)
    pool = [synth.rgbd_frame(seq, t, cfg) if workload == "rgbd" else synth.stereo_frame(seq, t, cfg) for t in range(16)]
    th = 15.0 if workload == "rgbd" else 7.0
    mk = lambda: orc.Extractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    ex = [(mk(), mk()) for _ in range(threads)]
    def work(k):
        frames = [pool[(k + t) % len(pool)] for t in range(frames_per_thread)]
        _cpu_walk(orc, workload, cfg, frames, ex[k][0], ex[k][1], cam10, I, th)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    return threads * frames_per_thread / dt, dt


# --------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="frames per step per GPU")
    ap.add_argument("--workload", choices=["rgbd", "stereo", "rgbd-cull", "rgbd-bow", "stereo-yolo"], default="rgbd")
    ap.add_argument("--cpu-frames", type=int, default=320, help="frames of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with hipEvents")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the per-GPU batch over this many HIP streams (latency-bound kernels of one stream overlap "
                         "with throughput-bound kernels of another)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    if os.environ.get("SD_BENCH_SINGLE_DEVICE"):     # rehearsal only: several ranks share cuda:0 (gloo backend)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
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

    pkg = graft.load_package()
    fe, synth = pkg.frontend, pkg.synth
    if fe.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")

    B = args.batch
    S = max(1, args.streams)
    if B % S:
        raise SystemExit("--batch must be a multiple of --streams")
    Bs = B // S
    cull = args.workload == "rgbd-cull"
    bow = args.workload == "rgbd-bow"
    yolo_wl = args.workload == "stereo-yolo"
    if cull or bow:
        args.workload = "rgbd"
    if yolo_wl:
        args.workload = "stereo"
        if args.streams != 1:
            raise SystemExit("--workload stereo-yolo runs on one stream")
    if args.workload == "rgbd":
        cfg = synth.KITTI03_RGBD
        workload_name = "KITTI-03 RGB-D 1241x376, 2000 feat/frame, ORB extract+match, no semantic mask (BASELINE configs[1])"
        if cull:
            workload_name = ("RGB-D 1241x376, 2000 feat/frame, extract + match + dynamic cull with 3 given boxes/frame "
                             "(firstSeparate, Separate vs the frame 0.2 s back, UpdateFrame); detector off (BASELINE configs[3] data path)")
        if bow:
            workload_name = ("RGB-D 1241x376, 2000 feat/frame, extract + projection match + Frame::ComputeBoW (k=10, L=6 vocabulary) + "
                             "SearchByBoW against the previous frame (TrackReferenceKeyFrame's matcher)")
        imgs_per_frame, th = 1, 15.0
    else:
        cfg = synth.KITTI_STEREO
        workload_name = "KITTI stereo 1241x376, 2000 feat/frame, 2x ORB extract + stereo match + projection match, detector off (BASELINE configs[2] minus YOLOv3)"
        if yolo_wl:
            workload_name = ("KITTI stereo 1241x376, 2000 feat/frame: YOLOv3 (640x480, synthetic weights) on the left image -> boxes -> boxTrack -> "
                             "2x ORB extract + stereo match + firstSeparate + TrackHomo (projection match vs the frame 0.2 s back, H/F fit) + "
                             "Separate + UpdateFrame + projection match vs the last frame (BASELINE configs[2])")
        imgs_per_frame, th = 2, 7.0
    W, H = cfg["width"], cfg["height"]

    # ---- synthetic inputs, resident in HBM before the timed region
    seq = 10 + rank
    if args.workload == "rgbd":
        fr = [synth.rgbd_frame(seq + 100 * (t // 128), t % 128, cfg) for t in range(B)]      # a synthetic sequence is valid for ~128 frames (zoom 1.01^t)
        d_rgb = torch.from_numpy(np.stack([f[0] for f in fr])).to(dev)
        d_depth = torch.from_numpy(np.stack([f[1] for f in fr]).view(np.int16)).to(dev)
        Wg = (W + 63) // 64 * 64                        # the gray plane is this pipeline's own intermediate: 64-byte aligned rows
        d_gray = torch.empty((B, H, Wg), dtype=torch.uint8, device=dev)
    else:
        fr = [synth.stereo_frame(seq + 100 * (t // 128), t % 128, cfg) for t in range(B)]
        Wg = W                                           # stereo inputs arrive as tight 8-bit images
        d_gray = torch.from_numpy(np.stack([im for f in fr for im in (f[0], f[1])])).to(dev)
    del fr

    # ---- start-up collective: the packed ORB vocabulary (synthetic k=10, L=6 tree in ORBvoc.txt's shape; the real file
    # is a download that never was in the reference), rank 0 -> all over RCCL; every rank adopts the received buffer
    # (sd_vocab_from_packed_device) and Frame::ComputeBoW / SearchByBoW read it from HBM.
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

    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    n_img = Bs * imgs_per_frame                     # images per stream
    cam = fe.make_camera(cfg)
    main_stream = torch.cuda.current_stream()
    streams = [main_stream] + [torch.cuda.Stream(device=dev) for _ in range(S - 1)]
    batches = [fe.Batch(ex, W, H, n_img + 1) for _ in range(S)]     # + 1 slot: mLastFrame carried across steps
    batch = batches[0]
    I = np.tile(np.eye(4, dtype=np.float32), (Bs, 1, 1))
    cur_idx = np.arange(Bs, dtype=np.int32) * imgs_per_frame
    last_idx = np.concatenate([[n_img], cur_idx[:-1]]).astype(np.int32)
    depth_factor = float(np.float32(1.0) / np.float32(cfg.get("depth_map_factor", 1.0)))
    cull_state = None
    if cull:
        # detector boxes per frame: precomputed (boxTrack is host code on a per-sequence recurrence; ids are stable here)
        bl, il = [], []
        for t in range(B):
            rows = synth.boxes_for_frame(seq, t, cfg)
            bl.append(synth.rows_to_rects(rows)); il.append(np.arange(len(rows), dtype=np.int32))
        dt = 2                                                    # reference frame = 0.2 s older at 10 fps
        sc = 1.01 ** dt
        Hm = np.array([[sc, 0, (3.0 * dt - cfg["cx"]) * sc + cfg["cx"]], [0, sc, -cfg["cy"] * sc + cfg["cy"]], [0, 0, 1]], np.float32)
        cull_state = []
        for k in range(S):
            f0 = k * Bs
            packed = fe.Batch.pack_boxes(bl[f0:f0 + Bs], il[f0:f0 + Bs])
            npair = Bs - dt
            li = np.zeros((npair, fe.MAXB), np.int32); ls = np.full((npair, fe.MAXB), -1, np.int32); nl = np.full(npair, 3, np.int32)
            li[:, :3] = np.arange(3)
            cull_state.append(dict(packed=packed, slots=np.arange(Bs, dtype=np.int32), cur=np.arange(dt, Bs, dtype=np.int32),
                                   ref=np.arange(0, Bs - dt, dtype=np.int32), H=np.tile(Hm.reshape(1, 9), (npair, 1)),
                                   flag=np.ones(npair, np.int32), last=(li, ls, nl)))
    det = None
    if yolo_wl:
        layers, anchors = pkg.yolo.v3_layers()
        det = pkg.yolo.Detector(layers, anchors, 640, 480, max_batch=B)
        det.load_weights(pkg.yolo.synth_weights(layers, seed=3)[0])
        d_bgr = d_gray.view(B, 2, H, W)[:, 0].unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()      # the left image as 3 channels
        yolo_state = dict(last=None, n_boxes=0, in_flight=False)
        det_stream = torch.cuda.Stream(device=dev)
    recs, gatherer = None, None
    if dist is not None:
        recs = []
        for bt in batches:
            kp_p, desc_p, cnt_p, cap = bt.results_device()
            recs += [fe.as_torch_u8(kp_p, n_img * cap * 28), fe.as_torch_u8(desc_p, n_img * cap * 32)]
        gatherer = ResultGather(dist, world, rank, sum(r.numel() for r in recs), dev)

    def run_yolo_step(first):
        """BASELINE configs[2], one batch of B consecutive stereo frames (frame i's reference frame is frame i-2, 0.2 s back)."""
        bt, st = batch, main_stream.cuda_stream
        # The detector (MFMA-bound) runs on its own stream, one batch AHEAD of the front end (VALU / latency-bound kernels + the
        # host's boxTrack recurrence): the boxes of this batch were requested during the previous step, the next batch's
        # forward pass is launched as soon as they are downloaded, and everything below overlaps with it.
        if not yolo_state["in_flight"]:
            det.forward_device(d_bgr.data_ptr(), W, H, W * 3, W * H * 3, B, 0.5, det_stream.cuda_stream)  # yolo->Segmentation_(imLeft)
        dets = det.boxes_batch(B, W, H, stream=det_stream.cuda_stream)                      # one synchronisation per batch
        det.forward_device(d_bgr.data_ptr(), W, H, W * 3, W * H * 3, B, 0.5, det_stream.cuda_stream)      # the next batch
        yolo_state["in_flight"] = True
        bt.extract_device(d_gray.data_ptr(), W, W * H, n_img, st)
        bt.stereo_match(B, cfg["bf"], cfg["fx"], st)
        # Frame::boxTrack is a host recurrence over the sequence (f64, a handful of boxes)
        bl, il = [], []
        lo, li_, lm, lv = np.zeros((0, 4)), np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros((0, 2))
        for i in range(B):
            bx, idx, omit, vel = fe.box_track(dets[i][0][:fe.MAXB // 2], lo, li_, lm, lv, W, H)
            bl.append(bx); il.append(idx)
            lo, li_, lm, lv = bx, idx, omit, vel
        yolo_state["n_boxes"] = int(np.mean([len(x) for x in bl]))
        bt.first_separate(cur_idx, None, None, stream=st, packed=fe.Batch.pack_boxes(bl, il))
        bt.assign_grid(n_img, cam, st)
        bt.unproject(imgs_per_frame, B, cam, I, st)
        # Tracking::TrackHomo against the frame 0.2 s back, then Separate / UpdateFrame
        bt.search_by_projection(cur_idx[2:], cur_idx[:-2], I[2:], I[2:], cam, th, False, True, stream=st)
        bt.estimate_motion(st)
        npair = B - 2
        lidx = np.zeros((npair, fe.MAXB), np.int32); lst = np.full((npair, fe.MAXB), -1, np.int32); nl = np.zeros(npair, np.int32)
        for p in range(npair):
            m = min(len(il[p + 1]), fe.MAXB); nl[p] = m; lidx[p, :m] = il[p + 1][:m]         # mLastFrame = frame p+1
        bt.separate(cur_idx[2:], cur_idx[:-2], None, None, None, None, stream=st, packed_last=(lidx, lst, nl))
        bt.update_frame(True, st)
        bt.assign_grid(n_img, cam, st)
        # TrackWithMotionModel's matcher against the last frame
        if first:
            bt.search_by_projection(cur_idx[1:], last_idx[1:], I[1:], I[1:], cam, th, False, True, stream=st)
        else:
            bt.search_by_projection(cur_idx, last_idx, I, I, cam, th, False, True, stream=st)
        bt.copy_frame(int(cur_idx[-1]), n_img, st)

    def run_stream(k, first):
        bt, st = batches[k], streams[k].cuda_stream
        f0 = k * Bs
        g0 = d_gray[f0 * imgs_per_frame:]
        if args.workload == "rgbd":            # GrabImageRGBD's cvtColor runs inside the level-0 copy of the extractor (same results as the two calls)
            bt.extract_color_device(d_rgb[f0:].data_ptr(), W * 3, W * H * 3, n_img, True, st)
        else:
            bt.extract_device(g0.data_ptr(), Wg, Wg * H, n_img, st)
        if args.workload == "rgbd":
            bt.rgbd_from_u16(d_depth[f0:].data_ptr(), W, W * H, Bs, depth_factor, cfg["bf"], st)
        else:
            bt.stereo_match(Bs, cfg["bf"], cfg["fx"], st)
        if cull:
            cs = cull_state[k]
            bt.first_separate(cs["slots"], None, None, stream=st, packed=cs["packed"])
        bt.assign_grid(n_img, cam, st)
        bt.unproject(imgs_per_frame, Bs, cam, I, st)
        if first:
            bt.search_by_projection(cur_idx[1:], last_idx[1:], I[1:], I[1:], cam, th, False, True, stream=st)
        else:
            bt.search_by_projection(cur_idx, last_idx, I, I, cam, th, False, True, stream=st)
        if cull:
            cs = cull_state[k]
            bt.separate(cs["cur"], cs["ref"], cs["H"], cs["flag"], None, None, stream=st, packed_last=cs["last"])
            bt.update_frame(True, st)
            bt.assign_grid(n_img, cam, st)                       # UpdateFeaturesToGrid
        if bow:
            bt.compute_bow(vocab, cur_idx, 4, st)
            bt.search_by_bow(cur_idx[:-1], cur_idx[1:], 0.7, True, stream=st)
        bt.copy_frame(int(cur_idx[-1]), n_img, st)

    def step(first=False):
        if yolo_wl:
            run_yolo_step(first)
            if dist is not None:
                gatherer.submit(recs)
            return
        for k in range(S):
            run_stream(k, first)
        if S > 1:                                   # join the side streams into the main one
            for k in range(1, S):
                main_stream.wait_stream(streams[k])
        if dist is not None:
            gatherer.submit(recs)
        if S > 1:                                   # next step's side-stream work must not overtake the gather
            for k in range(1, S):
                streams[k].wait_stream(main_stream)

    step(first=True)                      # priming: fills the carried mLastFrame slot (setup, untimed)
    for _ in range(args.warmup):
        step()
    for bt in batches:
        bt.sync()
    torch.cuda.synchronize()
    if not args.no_profile:
        for bt in batches:
            bt.set_profiling(True)
            bt.reset_kernel_times()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if gatherer is not None:
        gatherer.finish()                          # the last step's gather is inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(dist, elapsed, dev)
    for bt in batches:
        bt.sync()

    kt = {}
    if not args.no_profile:
        for bt in batches:
            for k, (ms, n) in bt.kernel_times().items():
                a, c = kt.get(k, (0.0, 0))
                kt[k] = (a + ms, c + n)
    counts = batch.counts(n_img)
    m, pairs, nm = batch.download_matches(Bs - 2 if bow else Bs - 1)      # rgbd-bow: the last SearchByBoW pair

    if rank == 0:
        total_frames = world * B * args.steps
        value = total_frames / elapsed
        alg = algorithmic_bytes(W, H, ex.mvInvScaleFactor, cfg["n_features"])
        roof = None
        if kt:
            dom = max((k for k in kt if kt[k][1] > 0), key=lambda k: kt[k][0])
            ms, launches = kt[dom]
            avg_ms = ms / launches
            per_launch = {"k_pyr_level": alg["k_pyr_level"] / 7.0}.get(dom, alg.get(dom, 0)) * n_img   # images per launch
            achieved = per_launch / (avg_ms * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    j = json.load(open(pmc))
                    e = j.get(dom) or j.get({"k_fast_cells": "k_fast_cells_staged", "k_blur": "k_blur_wide"}.get(dom, dom))      # in-library ids vs kernel symbols
                    if e and e.get("batch_images") == n_img:
                        traffic = e.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(per_launch), "avg_launch_ms": round(avg_ms, 4),
                    "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in kt.items() if v[1] > 0},
                    "streams": S,
                    "pipeline_achieved_GBs": round(alg["image_total"] * imgs_per_frame * value / 1e9, 2)}
        cpu = None
        if world == 1 and args.cpu_frames > 0:
            v, dt = cpu_baseline(args.workload, cfg, args.cpu_frames, seq)
            cpu = {"value": round(v, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": "%d frames (64 distinct, cycled) of the same synthetic workload through the CPU oracle (oracle/), "
                             "1 thread, %.1f s; host has %d logical cores" % (args.cpu_frames, dt, os.cpu_count() or 0)}
            nthr = max(1, min(32, (os.cpu_count() or 1) // 2))
            va, dta = cpu_baseline_all_cores(args.workload if args.workload in ("rgbd", "stereo") else "rgbd", cfg, seq, nthr)
            cpu["all_cores"] = {"value": round(va, 2), "unit": "frames/s", "threads": nthr,
                                "sample": "%d independent streams x 48 frames, %.1f s" % (nthr, dta)}
        out = {
            "metric": "tracking frames/sec (extract+match+dynamic-cull), KITTI 1241x376",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload_name, "frames_per_step_per_gpu": B, "images_per_frame": imgs_per_frame,
                       "features_per_image": int(np.mean(counts)), ("bow_matches_last_pair" if bow else "projection_matches_last_pair"): int(nm),
                       "sharding": "independent frame batches per rank, no data-path collective; per-step async gather of results to rank 0"
                       if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if yolo_wl:                               # the detector alone (the MFMA path), timed after the run
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(5):
                det.forward_device(d_bgr.data_ptr(), W, H, W * 3, W * H * 3, B, 0.5, main_stream.cuda_stream)
            torch.cuda.synchronize(); dt_det = (time.perf_counter() - t1) / 5
            fl = det.flops()
            out["detector"] = {"bound": "mfma", "images_per_s": round(B / dt_det, 1), "achieved": round(fl * B / dt_det / 1e12, 1), "peak": 2500.0,
                               "unit": "TFLOP/s", "frac": round(fl * B / dt_det / 2.5e15, 4), "gflop_per_image": round(fl / 1e9, 2),
                               "ms_per_batch": round(dt_det * 1e3, 3), "boxes_per_frame_after_boxTrack": yolo_state["n_boxes"],
                               "weights": "synthetic (yolov3.weights is a download that never was in the reference)"}
        # the broadcast vocabulary's consumer, outside the timed region: Frame::ComputeBoW of frame 0
        batch.compute_bow(vocab, [0], 4)
        b0 = batch.download_bow(0)
        out["vocabulary"] = {"nodes": vocab.info()["n_nodes"], "words": vocab.info()["n_words"], "packed_bytes": int(voc_bytes),
                             "frame0_bow_words": int(len(b0["word"])), "frame0_feature_vector_nodes": int(len(np.unique(b0["fv_node"])))}
        if voc_ms is not None:
            out["vocabulary_broadcast_ms"] = round(voc_ms, 3)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for bt in batches:
        bt.close()


if __name__ == "__main__":
    main()
