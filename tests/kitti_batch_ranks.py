"""TEST INFRASTRUCTURE (run by tests/test_gpu_headline.py, never by the product): bench.SequenceBatchWorkload -- BASELINE configs[4] as bench.py
times it -- on one or several ranks, every frame of every lane checked against the oracle chain (torch-fp32 YOLOv3 forward -> region decode ->
postprocess_ -> SequenceOracle(rgb_order=True).track; /root/reference Examples/Stereo/stereo_kitti.cc:81-155).  Several ranks = processes that share
cuda:0 over gloo (SD_BENCH_SINGLE_DEVICE / SD_BENCH_BACKEND=gloo, the rehearsal mode of bench.py): what is rehearsed is the frame hand-over
(worker -> records -> all-to-all -> owner's prefetched block), not RCCL.

    python tests/kitti_batch_ranks.py --gpus N --sequences Q --frames T --block-frames B [--no-detector]
exit code 0 = every frame of every owned sequence equal to the oracle's on every rank."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--sequences", type=int, default=2)
    ap.add_argument("--frames", type=int, default=7)
    ap.add_argument("--block-frames", type=int, default=6)
    ap.add_argument("--no-detector", action="store_true")
    a = ap.parse_args()
    if os.environ.get("WORLD_SIZE") is None and a.gpus > 1:
        os.environ.update(SD_BENCH_SINGLE_DEVICE="1", SD_BENCH_BACKEND="gloo")
        sys.exit(bench.spawn_ranks(a.gpus, sys.argv[1:], script=os.path.abspath(__file__)))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        import datetime
        dist_mod.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
        dist = dist_mod
    pkg = graft.load_package()
    fe, synth = pkg.frontend, pkg.synth
    orc = graft.load_oracle()
    P = graft.load_pipeline_oracle()
    yo = graft.load_yolo_oracle()
    from test_gpu_pipeline import _check_frame
    from test_gpu_headline import _oracle_boxes
    cfg = synth.KITTI_STEREO
    W, H = cfg["width"], cfg["height"]
    args = argparse.Namespace(kitti_frames=a.frames, kitti_sequences=a.sequences, block_frames=a.block_frames)
    wl = bench.SequenceBatchWorkload(args, rank, world, dev, pkg, dist, detector=not a.no_detector)
    layers, anchors = pkg.yolo.v3_layers()
    per = pkg.yolo.synth_weights(layers, seed=3)[1] if not a.no_detector else None
    S = wl.S
    oracles = [P.SequenceOracle(orc, cfg, P.SENSOR_STEREO, rgb_order=True, track_last=True) for _ in range(S)]
    stats = dict(frames=0, most=0, flags=[], blocks=len(wl.plan["blocks"]), D=wl.D, computed=sum(len(v["mine"]) for v in wl.views))
    Pn = min(24, a.frames)

    def on_frame(t, res):
        for l in range(S):
            q = wl.my_sequences[l]
            fr = bench.synth_timestep(synth, "stereo", cfg, 10 + q, bench.pingpong_index(t, Pn))      # the frame prepare() parked for (q, t)
            left, right = fr["images"][0], fr["images"][1]
            boxes = fr["boxes"] if a.no_detector else _oracle_boxes(yo, orc, layers, anchors, per, left, W, H)
            stats["most"] = max(stats["most"], len(boxes))
            F = oracles[l].track(left, right, boxes, t / float(cfg["fps"]))
            _check_frame(fe, wl.trk, l, res[l], F, "rank %d frame %d lane %d (sequence %d)" % (rank, t, l, q))
            stats["frames"] += 1
            if F.ref_id >= 0:
                stats["flags"].append(F.track_flag)

    try:
        wl.prepare()
        wl.run(on_frame=on_frame)
        torch.cuda.synchronize()
        assert stats["frames"] == a.frames * S
        if wl.det is not None:
            assert wl.max_det_boxes == stats["most"], "the detector's largest box list %d vs the oracle's %d" % (wl.max_det_boxes, stats["most"])
        assert any(f != 0 for f in stats["flags"]), "TrackHomo must have run: %r" % stats["flags"]
    finally:
        wl.close()
    print("KITTI_BATCH_OK rank %d/%d lanes %r frames %d blocks %d D %d computed %d most_boxes %d flags %r"
          % (rank, world, wl.my_sequences, stats["frames"], stats["blocks"], stats["D"], stats["computed"], stats["most"], stats["flags"]), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)          # at once: a peer waiting in a collective sees the connection close instead of waiting for its timeout
