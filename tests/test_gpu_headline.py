"""GPU parity of the HEADLINE chain exactly as bench.py times it (BASELINE configs[2]; /root/reference Examples/Stereo/stereo_kitti.cc:107-122,
src/yolo.cc:60-77,151-206): 3-channel KITTI stereo pairs -> blobFromImage + YOLOv3 in f32 (k_conv_f32) + region decode + NMS on the device
(`forward_device` -> `boxes_device`, one frame ahead on the detector's stream, two result slots) -> every kept box -> sd_tracker_track with
channels = 3 (cvtColor fused into the level-0 copy) -> extract x2, stereo match, boxTrack, firstSeparate, TrackHomo, Separate, UpdateFrame,
match vs mLastFrame.  The test drives bench.Workload("stereo-yolo") itself -- the object the timed loop steps -- and compares every lane of
every frame bit for bit with the oracle chain: torch-fp32 forward -> region_decode -> postprocess_ -> SequenceOracle(rgb_order=True).track.
Frames with MORE THAN 16 detector boxes are part of it (round 2's bench cut the list there; nothing is cut any more)."""
import argparse
import os
import sys

import numpy as np
import pytest

import __graft_entry__ as graft

pytestmark = pytest.mark.gpu


def _oracle_boxes(yo, orc, layers, anchors, per, img, W, H):
    blob = yo.blob_from_image(img, 640, 480, orc.resize_linear)
    ref = yo.torch_forward(layers, per, blob)
    rows = np.concatenate([yo.region_decode(ref[li - 1][0].numpy().transpose(1, 2, 0), list(layers[li]["mask"]), anchors, 640, 480)
                           for li in (82, 94, 106)])
    return yo.postprocess(rows, W, H, 0.5, 0.4)[0]


def test_headline_chain_colour_stereo_f32_detector(gpu, fe, orc, synth, pkg):
    import torch
    sys.path.insert(0, graft.ROOT)
    import bench
    from test_gpu_pipeline import _check_frame, _pipe
    P = _pipe()
    yo = graft.load_yolo_oracle()
    cfg = synth.KITTI_STEREO
    W, H = cfg["width"], cfg["height"]
    S, T = 2, 5                        # 10 fps: the first frame more than 0.2 s older is three back, so TrackHomo runs on frames 3 and 4
    args = argparse.Namespace(lanes=S, distinct=S, det_split=1, kitti_frames=256)
    dev = torch.device("cuda", 0)
    wl = bench.Workload("stereo-yolo", args, 0, 1, dev, pkg, None)
    layers, anchors = pkg.yolo.v3_layers()
    _, per = pkg.yolo.synth_weights(layers, seed=3)
    oracles = [P.SequenceOracle(orc, cfg, P.SENSOR_STEREO, rgb_order=True, track_last=True) for _ in range(S)]
    most, flags, appended = 0, [], 0
    try:
        assert wl.det is not None and wl.det_prec == "f32" and wl.lookahead
        wl.prepare(T)
        for t in range(T):
            res = wl.step()
            for l in range(S):
                fr = bench.synth_timestep(synth, "stereo", cfg, 10 + l, t)          # what Workload.prepare parked in HBM for lane l (rank 0)
                left, right = fr["images"][0], fr["images"][1]
                boxes = _oracle_boxes(yo, orc, layers, anchors, per, left, W, H)
                most = max(most, len(boxes))
                F = oracles[l].track(left, right, boxes, fr["stamp"])
                _check_frame(fe, wl.trk, l, res[l], F, "frame %d lane %d" % (t, l))
                if F.ref_id >= 0:
                    flags.append(F.track_flag)
                appended += 0 if F.appended is None else len(F.appended)
        assert wl.max_det_boxes == most, "the detector's largest box list %d vs the oracle's %d" % (wl.max_det_boxes, most)
    finally:
        wl.close()
    assert most > 16, "the sequences must contain a frame with more than 16 detector boxes (had at most %d)" % most
    assert len(flags) >= 2 * S and any(f != 0 for f in flags), "TrackHomo must have run on frames 3 and 4: %r" % flags


def _run_ranks(argv, timeout=900):
    import subprocess
    script = os.path.join(graft.ROOT, "tests", "kitti_batch_ranks.py")
    p = subprocess.run([sys.executable, script] + argv, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, "kitti_batch_ranks.py %s failed:\n%s\n%s" % (" ".join(argv), p.stdout[-3000:], p.stderr[-3000:])
    return [l for l in p.stdout.splitlines() if l.startswith("KITTI_BATCH_OK")]


def test_kitti_batch_as_benched_f32_detector(gpu):
    """BASELINE configs[4] exactly as bench.py times it at N = 1 (bench.SequenceBatchWorkload.run; /root/reference Examples/Stereo/stereo_kitti.cc:81-155):
    two 3-channel stereo sequences x 7 frames, the f32 detector over blocks of up to D = 3 frames per sequence, the history-free half through
    sd_tracker_prefetch -> export -> import, boxes handed over behind the records, four blocks (1 + 3 + 2 + 1 frames, bench.block_schedule: an
    outstanding block is consumed while the next is in flight).  Every frame of every lane == torch-fp32 boxes -> SequenceOracle."""
    ok = _run_ranks(["--gpus", "1", "--sequences", "2", "--frames", "7", "--block-frames", "6"])
    assert len(ok) == 1 and "blocks 4 D 3" in ok[0] and "frames 14" in ok[0], ok


def test_kitti_batch_frames_sharded_over_two_ranks(gpu):
    """configs[4] on two ranks (processes sharing the one GPU, gloo): three sequences -- rank 0 owns two, rank 1 one (its second lane repeats it) --,
    every time block's frames dealt evenly to the two ranks whoever owns them, records + boxes through the all-to-all; every frame on every rank
    equals the sequential oracle.  Given boxes (no detector: two f32 detectors' activations per process are not what this test is about) and
    once with the detector on a shorter run."""
    ok = _run_ranks(["--gpus", "2", "--sequences", "3", "--frames", "7", "--block-frames", "5", "--no-detector"])
    assert len(ok) == 1 or len(ok) == 2, ok          # rank 0's line is always passed through
    ok = _run_ranks(["--gpus", "2", "--sequences", "3", "--frames", "5", "--block-frames", "4"])
    assert ok, ok


@pytest.mark.parametrize("name", ["stereo", "rgbd-cull"])
def test_pipelined_detectorless_workloads_as_benched(gpu, fe, orc, synth, pkg, name):
    """bench.Workload for the detector-less workloads as bench.py steps them since round 4: the history-free half of frame t + 1 (sd_tracker_prefetch,
    look-ahead 1) is enqueued before frame t's call synchronises.  Two lanes x 6 frames of `stereo` (colour pairs, no boxes) and `rgbd-cull` (colour +
    depth, three given boxes per frame: the cull runs on the prefetched frames): every lane of every frame equals the sequential frame-level oracle."""
    import torch
    sys.path.insert(0, graft.ROOT)
    import bench
    from test_gpu_pipeline import _check_frame, _pipe
    P = _pipe()
    S, T = 2, 6
    args = argparse.Namespace(lanes=S, distinct=S, det_split=1, kitti_frames=256)
    wl = bench.Workload(name, args, 0, 1, torch.device("cuda", 0), pkg, None)
    cfg = wl.cfg
    sensor = P.SENSOR_STEREO if wl.kind == "stereo" else P.SENSOR_RGBD
    oracles = [P.SequenceOracle(orc, cfg, sensor, rgb_order=True, track_last=True) for _ in range(S)]
    flags = 0
    try:
        assert wl.pipelined and wl.det is None
        wl.prepare(T)
        for t in range(T):
            res = wl.step()
            for l in range(S):
                fr = bench.synth_timestep(synth, wl.kind, cfg, 10 + l, t)
                second = fr["images"][1] if wl.kind == "stereo" else fr["depth"]
                F = oracles[l].track(fr["images"][0], second, fr["boxes"] if wl.with_boxes else None, fr["stamp"])
                _check_frame(fe, wl.trk, l, res[l], F, "%s frame %d lane %d" % (name, t, l))
                flags += F.track_flag != 0
    finally:
        wl.close()
    assert name == "stereo" or flags >= 2, "TrackHomo must have run on the later frames of the cull workload"
