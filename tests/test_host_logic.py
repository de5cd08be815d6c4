"""Host logic of bench.py and the synthetic harness (no GPU): sharding, byte accounting, gloo collectives."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_match_survey(fe):
    ex = fe.ORBextractor(2000, 1.2, 8, 20, 7)
    b = bench.algorithmic_bytes(1241, 376, ex.mvInvScaleFactor, 2000)
    # 466,616 + 1,738,559 + 1,407,767 + 3*1,444,097 + 1,498,000 + 1,024,000 + 120,000
    assert b["image_total"] == 10587233
    assert b["k_fast_cells"] == 1444097 and b["k_blur"] == 2 * 1444097
    assert b["k_pyr_level0"] + b["k_pyr_level"] == 466616 + 1738559 + 1407767
    ex = fe.ORBextractor(1000, 1.2, 8, 20, 7)
    assert bench.algorithmic_bytes(640, 480, ex.mvInvScaleFactor, 1000)["image_total"] == 6564354


def test_shard_sequences_longest_first():
    lengths = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]     # KITTI 00-10
    owner = bench.shard_sequences(11, lengths, 8)
    assert sorted(set(owner)) == list(range(8))
    load = [sum(l for l, o in zip(lengths, owner) if o == r) for r in range(8)]
    assert max(load) == 4661                                       # the longest sequence bounds the makespan
    assert bench.shard_sequences(3, [5, 5, 5], 1) == [0, 0, 0]


def test_synth_is_deterministic(synth):
    a = synth.stereo_frame(seq=3, t=2)
    b = synth.stereo_frame(seq=3, t=2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    rgb, depth, ts = synth.rgbd_frame(seq=3, t=1)
    assert rgb.shape == (376, 1241, 3) and depth.dtype == np.uint16 and abs(ts - 0.1) < 1e-9
    assert depth.min() > 0
    rows = synth.boxes_for_frame(3, 0)
    rects = synth.rows_to_rects(rows)
    assert rects.shape == (3, 4) and (rects[:, 2] >= 60).all()
    m = synth.mask_from_boxes(rows, 1241, 376)
    assert set(np.unique(m).tolist()) <= {0, 255} and (m == 255).any()


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = torch.full((5, 7), rank + 1, dtype=torch.uint8)
    g = bench.gather_records(dist, local, world)
    t = bench.max_over_ranks(dist, 1.0 + rank, torch.device("cpu"))
    rg = bench.ResultGather(dist, world, rank, 35, torch.device("cpu"))
    for step in range(3):                      # async gather to rank 0, re-submitted while the previous one may be in flight
        rg.submit([torch.full((5, 7), 10 * step + rank, dtype=torch.uint8).reshape(-1)])
    rg.finish()
    if rank == 0:
        assert [int(r[0]) for r in rg.recv] == [20, 21]
    voc = torch.arange(100, dtype=torch.uint8) if rank == 0 else torch.zeros(100, dtype=torch.uint8)
    dist.broadcast(voc, src=0)
    q.put((rank, g.shape, [int(g[r].float().mean()) for r in range(world)], t, int(voc.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_rank_gather_and_timing_gloo():
    """world_size 2 over gloo on CPU: the N>1 collectives bench.py uses (all_gather of result records,
    MAX of rank times, vocabulary broadcast)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, shape, means, t, vsum in res:
        assert tuple(shape) == (2, 5, 7) and means == [1, 2] and t == 2.0 and vsum == sum(range(100))


def test_result_record_layout_roundtrip(fe):
    """The per-frame record a rank gathers to rank 0 (N, keypoints, descriptors, uRight, depth, boxes / box_status, N_s)."""
    cap = 2064
    lay = bench.record_layout(cap, fe.FRAME_BOXES_BYTES)
    M = fe.MAXB                                   # SD_MAX_BOXES
    FB = 16 + M * 32 + 3 * M * 4 + (M + 1) * 4 + 4    # sizeof(sd_frame_boxes)
    assert lay["_stride"] % 16 == 0 and lay["kp"][1] == cap * 28 and lay["fb"][1] == FB == 3096
    rec = np.zeros(lay["_stride"], np.uint8)
    N, Ns = 1800, 1750
    rec[lay["count"][0]:lay["count"][0] + 4] = np.array([N], np.int32).view(np.uint8)
    fb = np.zeros(FB, np.uint8)
    fb[:16] = np.array([2, 2000, Ns, 250], np.int32).view(np.uint8)
    fb[16 + 32 * M:16 + 32 * M + 8] = np.array([7, 9], np.int32).view(np.uint8)
    fb[16 + 32 * M + 4 * M:16 + 32 * M + 4 * M + 8] = np.array([2, -1], np.int32).view(np.uint8)
    rec[lay["fb"][0]:lay["fb"][0] + FB] = fb
    ur = np.arange(cap, dtype=np.float32)
    rec[lay["uright"][0]:lay["uright"][0] + cap * 4] = ur.view(np.uint8)
    d = bench.decode_record(rec, lay, cap)
    assert (d["N"], d["N_s"], d["N_d"], d["n_boxes"]) == (N, Ns, 250, 2)
    assert d["box_idx"].tolist() == [7, 9] and d["box_status"].tolist() == [2, -1]
    assert np.array_equal(d["uright"], ur[:N]) and int(d["readmitted"].sum()) == N - Ns and d["desc"].shape == (N, 32)


_RANK_SCRIPT = """
import json, os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([rank + 1.0])
dist.all_reduce(t)
if rank == 0:
    print(json.dumps({"rccl_ranks": dist.get_world_size(), "sum": float(t.item()), "local_rank": int(os.environ["LOCAL_RANK"])}))
dist.barrier(); dist.destroy_process_group()
sys.exit(int(sys.argv[1]) if rank == 1 else 0)
"""


def test_self_launcher_starts_all_ranks(tmp_path, capfd):
    """`python bench.py --gpus N` without a torch.distributed environment: bench.spawn_ranks starts N ranks (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), rank 0's JSON line passes through, any failing rank fails the run."""
    import json
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    assert bench.spawn_ranks(2, ["0"], script=str(script)) == 0
    out = capfd.readouterr().out.strip().splitlines()
    j = json.loads(out[-1])
    assert j == {"rccl_ranks": 2, "sum": 3.0, "local_rank": 0}
    assert bench.spawn_ranks(2, ["3"], script=str(script)) != 0


def test_oracle_local_map_best_second_ratio():
    """ORBmatcher::SearchByProjection(Frame, MapPoints) (ORBmatcher.cc:45-129) on a hand-made case: the ratio test only
    applies when best and second best share a level; a keypoint taken by an earlier point with observations is skipped."""
    import numpy as np
    import __graft_entry__ as g
    orc = g.load_oracle()
    KP = orc.KP_DTYPE
    cam10 = np.array([500, 500, 320, 240, 40, 0.08, 0, 640, 0, 480], np.float32)
    sf = np.array([1.2 ** l for l in range(8)], np.float32)
    kp = np.zeros(3, KP)
    kp["x"] = [320, 322, 318]; kp["y"] = [240, 241, 239]; kp["octave"] = [0, 0, 1]
    desc = np.zeros((3, 32), np.uint8)
    desc[1, 0] = 0xF0          # 4 bits from the map point descriptor (all zero)
    desc[2, :2] = 0xFF         # 16 bits
    desc[0, 0] = 0x0F          # 4 bits, visited first (same cell, lower index)
    ur = -np.ones(3, np.float32)
    mp = np.zeros(2, orc.MAP_POINT_DTYPE)
    mp["xw"] = [[0, 0, 5], [0, 0, 5]]; mp["normal"] = [[0, 0, 1], [0, 0, 1]]
    mp["min_distance"] = 1; mp["max_distance"] = 5.5; mp["flags"] = [3, 3]
    md = np.zeros((2, 32), np.uint8)
    T = np.eye(4, dtype=np.float32)
    track, mpm, kpm, nm = orc.search_local_map(kp, desc, ur, mp, md, T, cam10, sf, 1.0, 0.8)
    assert track["in_view"].tolist() == [1, 1] and track["level"].tolist() == [1, 1]
    assert track["proj_x"][0] == 320 and track["proj_y"][0] == 240 and track["view_cos"][0] == 1.0
    # point 0: best = kp0 (4 bits, level 0), second = kp1 (4 bits, level 0): 4 > 0.8*4 -> rejected by the ratio test
    # point 1: same candidates, same outcome
    assert mpm.tolist() == [-1, -1] and nm == 0
    desc[1, 0] = 0xFF          # second best now 8 bits: 4 <= 6.4 -> accepted; point 1 then sees kp0 taken:
    track, mpm, kpm, nm = orc.search_local_map(kp, desc, ur, mp, md, T, cam10, sf, 1.0, 0.8)
    # its best is kp1 (8 bits, level 0), second kp2 (16 bits, level 1): levels differ -> no ratio test
    assert mpm.tolist() == [0, 1] and kpm.tolist() == [0, 1, -1] and nm == 2


def test_bench_helpers_cpu_share_and_detector_bytes():
    """bench.usable_cpus() never exceeds the affinity mask and is at least 1; detector_algorithmic_bytes() of the built-in YOLOv3 list at
    640 x 480: every convolution's input + output (+ the fused shortcut operand) once per image, the weights once per launch."""
    import os
    import bench
    import __graft_entry__ as graft
    n = bench.usable_cpus()
    assert 1 <= n <= (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    pkg = graft.load_package()
    layers, _ = pkg.yolo.v3_layers()
    one = bench.detector_algorithmic_bytes(layers, 640, 480, 1)
    two = bench.detector_algorithmic_bytes(layers, 640, 480, 2)
    weights = 2 * one - two                                  # the batch-independent part
    cins = pkg.yolo.conv_inputs(layers)
    expect = sum(int(l["filters"]) * cins[i] * int(l["size"]) ** 2 for i, l in enumerate(layers) if l["type"] == pkg.yolo.CONV)
    assert weights == 4 * expect == 4 * 61_895_776           # yolov3.cfg's convolution weights (its 62,001,757 parameters minus biases / batch-norm terms), f32
    assert (one - weights) % 4 == 0 and 600e6 < one - weights < 700e6        # ~669 MB of activation traffic per image
    assert bench.detector_algorithmic_bytes(layers, 640, 480, 1, elt=2) * 2 == one


def test_bench_pingpong_index_keeps_neighbours():
    """Long runs walk the resident time steps back and forth: the first P steps are the resident ones in order, and any two consecutive
    steps show resident steps that are neighbours (|difference| == 1), so the tracker always sees consecutive views of one scene."""
    import bench
    for P in (2, 3, 5, 48):
        seq = [bench.pingpong_index(t, P) for t in range(5 * P)]
        assert seq[:P] == list(range(P))
        assert all(0 <= i < P for i in seq)
        assert all(abs(a - b) == 1 for a, b in zip(seq, seq[1:]))
    assert [bench.pingpong_index(t, 1) for t in range(4)] == [0, 0, 0, 0]


def test_sequence_batch_block_plan():
    """configs[4]: whole sequences per rank (11 over 8 ranks: at most 2 per rank), and blocks of D consecutive frames per owned sequence so that one
    detector / extraction batch holds >= 64 frames whatever the rank count."""
    for world in (1, 2, 4, 8):
        owner = bench.shard_sequences(11, [256] * 11, world)
        per_rank = [sum(1 for o in owner if o == r) for r in range(world)]
        assert sum(per_rank) == 11 and max(per_rank) - min(per_rank) <= 1
        S = max(per_rank)
        D, plan = bench.block_plan(256, S, 64, 24)
        assert S * D >= 64 and S * (D - 1) < 64
        assert sum(n for _, n, _ in plan) == 256 and plan[0][0] == 0 and all(len(idx) == n for _, n, idx in plan)
        flat = [i for _, _, idx in plan for i in idx]
        assert all(abs(a - b) == 1 for a, b in zip(flat, flat[1:])) and max(flat) == 23 and min(flat) == 0      # consecutive frames stay neighbouring views
    D, plan = bench.block_plan(5, 2, 64, 24)
    assert D == 5 and plan == [(0, 5, [0, 1, 2, 3, 4])]


def _canned_full_record(n_extras=12, kernels=16):
    """A full bench record at least as heavy as round 3's 24 KB one (per-kernel tables, the extras' rooflines, long sample texts)."""
    table = {"k_%02d" % i: {"alg_bytes_per_step": 987829248 + i, "ms_per_step": 0.336, "launches_per_step": 1.0, "achieved_GBs": 2940.4, "frac": 0.3675,
                             "traffic_bytes_per_step": 1056616106, "traffic_over_alg": 1.07, "traffic_profile": "r04_headline_pmc.csv"} for i in range(kernels)}
    fe_roof = {"bound": "hbm", "kernel": "k_fast_cells", "achieved": 463.84, "peak": 8000.0, "unit": "GB/s", "frac": 0.05798, "traffic": 801017941,
               "traffic_profile": "r04_headline_pmc.csv", "algorithmic_bytes_per_launch": 739377664, "avg_launch_ms": 1.594, "images_per_launch": 512,
               "kernels": table, "kernels_ms_per_step": {k: 0.3 for k in table}, "measured": "x" * 300, "note": "y" * 400}
    roof = {"bound": "mfma", "kernel": "k_conv_f32 x 75 launches = the convolutions of one 256-image detector batch", "achieved": 128.3, "peak": 157.3,
            "unit": "TFLOP/s", "frac": 0.8154, "limb_mode": None, "traffic": 364475262150, "traffic_profile": "r04_headline_pmc.csv",
            "algorithmic_flops_per_launch": 29931130060800, "avg_launch_ms": 233.371, "algorithmic_bytes_per_launch": 171526021504, "operands": "f32",
            "batch": 256, "measured": "m" * 300, "weights": "w" * 100, "front_end": fe_roof}
    extras = {"extra-%d" % i: {"value": 1000.0 + i, "unit": "frames/s", "ms_per_step": 1.0, "workload": "t" * 500, "roofline": fe_roof} for i in range(n_extras)}
    extras["broken"] = {"error": "RuntimeError: " + "e" * 300}
    return {"metric": "tracking frames/sec (extract+match+dynamic-cull), KITTI 1241x376", "value": 1067.57, "unit": "frames/s", "n_gpus": 1, "rccl_ranks": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 239.7978, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8+f32",
            "data": "synthetic", "config": {"workload": bench.WORKLOAD_TEXT["stereo-yolo"] + "; " + "q" * 400, "lanes_per_gpu": 256, "images_per_frame": 2,
                                             "frames_timed": 5120, "timed_seconds": 4.8, "detector_arithmetic": "f32 operands, f32 accumulation (v_mfma_f32_32x32x2_f32)",
                                             "sharding": "single GPU", "lane0_last_frame": {"N": 2000}},
            "roofline": roof, "cpu_baseline": {"value": 9.997, "unit": "frames/s", "cores": 16, "kind": "port", "sample": "s" * 900,
                                               "front_end_ms": {"median": 33.4}, "all_cores": {"value": 206.66, "threads": 16, "sample": "z" * 100}},
            "extra": extras, "vocabulary": {"nodes": 1049641}, "value_f32x3": 1566.08, "dtype_f32x3": "u8+f32 operands as 3xbf16 limbs, f32 accumulate"}


def test_bench_line_is_compact_and_complete(tmp_path):
    """The driver keeps only a few KB of stdout: the printed line must parse, stay below 4 KB whatever the detail record holds, and still
    carry metric / value / config.workload / roofline / cpu_baseline and one number per extra (round 3's 24 KB line was lost)."""
    full = _canned_full_record()
    assert len(json.dumps(full)) > 24000
    line = bench.compact_line(full, "bench_detail.json")
    assert "\n" not in line and len(line) < 4096
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert j[k] == full[k], k
    assert j["config"]["workload"].startswith("KITTI stereo 1241x376") and j["config"]["lanes_per_gpu"] == 256
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["frac"] == 0.8154 and r["achieved"] == 128.3 and r["peak"] == 157.3 and r["unit"] == "TFLOP/s"
    assert r["traffic"] == 364475262150 and r["avg_launch_ms"] == 233.371 and "kernels" not in r and "kernels" not in r["front_end"]
    assert r["front_end"]["kernel"] == "k_fast_cells" and r["front_end"]["bound"] == "hbm"
    c = j["cpu_baseline"]
    assert (c["value"], c["cores"], c["kind"], c["unit"]) == (9.997, 16, "port", "frames/s") and len(c["sample"]) <= 420
    assert j["extra"]["extra-3"] == 1003.0 and "error" in j["extra"]["broken"]
    assert j["value_f32x3"] == 1566.08 and j["detail"] == "bench_detail.json"
    # a record with no roofline / cpu baseline / extras (N > 1 ranks) still gives a valid line with the contract's keys
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")}
    j2 = json.loads(bench.compact_line(bare))
    assert j2["roofline"] is None and j2["cpu_baseline"] is None and j2["extra"] == {}
    # a pathological record (hundreds of extras) is cut down rather than printed over the limit
    huge = _canned_full_record(n_extras=400)
    l3 = bench.compact_line(huge, "bench_detail.json")
    assert len(l3) < 4096 and json.loads(l3)["roofline"]["frac"] == 0.8154
    # the detail record is what compact_line was given, on disk
    old_root = bench.ROOT
    try:
        bench.ROOT = str(tmp_path)
        os.makedirs(os.path.join(str(tmp_path), "gpurun_out"))
        assert bench.write_detail(full, "d.json") == "d.json"
        assert json.load(open(os.path.join(str(tmp_path), "gpurun_out", "d.json")))["extra"]["extra-0"]["roofline"]["kernels"]
    finally:
        bench.ROOT = old_root


def test_bench_auto_extras_are_the_baseline_configs():
    assert set(("rgbd", "tum-mask", "kitti-batch")) <= set(bench.AUTO_EXTRAS) and len(bench.AUTO_EXTRAS) <= 6
    assert all(w in bench.WORKLOADS for w in bench.AUTO_EXTRAS)


def _simulate_frame_shard(plan):
    """Run frame_shard_plan's hand-over on the host for every rank at once: each unit's row is its (sequence, absolute frame) pair.
    -> (units a rank computed over the whole job, per rank the [(t, lane sequences)] its tracker consumed in order)."""
    world = plan["world"]
    computed = [0] * world
    consumed = [[] for _ in range(world)]
    for B in plan["blocks"]:
        views = [bench.frame_shard_rank_view(plan, B, r) for r in range(world)]
        covered = sorted(u for v in views for u in v["mine"])
        assert covered == sorted(B["units"]) and len(set(covered)) == len(covered)             # every frame of the block is computed exactly once
        sent = []
        for r, v in enumerate(views):
            computed[r] += len(v["mine"])
            rows = [v["mine"][i] for i in v["send_perm"]]                                      # grouped by destination
            assert sum(v["send_splits"]) == len(rows)
            cut, o = [], 0
            for d in range(world):
                cut.append(rows[o:o + v["send_splits"][d]]); o += v["send_splits"][d]
                assert all(plan["owner"][q] == d for q, _ in cut[-1])
            sent.append(cut)
        for r, v in enumerate(views):
            recv = [u for src in range(world) for u in sent[src][r]]                           # what an all-to-all delivers, source by source
            assert [len(sent[src][r]) for src in range(world)] == v["recv_splits"] and recv == v["recv_units"]
            pool = [recv[i] for i in v["pool_index"]]
            S = plan["S"]
            for k in range(B["n"]):
                assert pool[k * S:(k + 1) * S] == [(q, k) for q in plan["lanes"][r]]
                consumed[r].append((B["t0"] + k, [q for q, _ in pool[k * S:(k + 1) * S]]))
    return computed, consumed


def test_frame_shard_plan_balances_frames_and_delivers_every_frame_to_its_owner():
    """configs[4], north star: "independent frames shard across the 8 GPUs".  The history-free half of the 11 x 256 frames is spread evenly over the
    ranks (round 3 sharded whole sequences: 2 : 1 at 8 ranks), every frame reaches the owner of its sequence, and an owner's tracker consumes its
    sequences' frames in time order."""
    for world in (1, 2, 3, 4, 8):
        plan = bench.frame_shard_plan(11, 256, world, 128, 24)
        computed, consumed = _simulate_frame_shard(plan)
        assert sum(computed) == 11 * 256
        mean = sum(computed) / world
        assert max(computed) / mean - 1 <= 0.05, (world, computed)                             # history-free load imbalance <= 5 %
        full = [B for B in plan["blocks"] if B["n"] >= plan["D"] - 1]
        per_block = [len(bench.frame_shard_rank_view(plan, B, r)["mine"]) for B in full for r in range(world)]
        assert full and min(per_block) >= 110                                                  # a full block's detector batch is about --block-frames images
        sizes = [B["n"] for B in plan["blocks"]]
        assert sum(sizes) == 256 and sizes[0] <= max(1, plan["D"] // 8) and sizes[-1] <= max(1, plan["D"] // 8)      # short first / last blocks: fill and drain
        owned = sorted(q for r in range(world) for q in set(plan["lanes"][r]))
        assert owned == list(range(11))
        for r in range(world):
            assert [t for t, _ in consumed[r]] == list(range(256))
            assert all(seqs == plan["lanes"][r] for _, seqs in consumed[r])
        for B in plan["blocks"]:
            assert all(abs(a - b) == 1 for a, b in zip(B["idx"], B["idx"][1:]))                # consecutive frames stay neighbouring views
    small = bench.frame_shard_plan(11, 7, 2, 16, 24)                                           # a short job: ramp, one full block, a ragged one, ramp
    assert small["D"] == 3 and [B["n"] for B in small["blocks"]] == [1, 3, 2, 1]
    _simulate_frame_shard(small)
    assert bench.block_schedule(256, 94) == [11, 23, 47, 94, 47, 23, 11] and bench.block_schedule(20, 12) == [1, 3, 6, 6, 3, 1]
    assert bench.block_schedule(256, 187) == [11, 23, 46, 48, 48, 46, 23, 11]                  # blocks too long for the job: halved until the ramps fit
    assert all(sum(bench.block_schedule(T, D)) == T for T in (1, 5, 7, 64, 256, 1000) for D in (1, 2, 3, 12, 47, 94, 300))
    with pytest.raises(ValueError):
        bench.frame_shard_plan(11, 256, 12, 128)


def _frame_shard_gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = bench.frame_shard_plan(11, 7, world, 16, 24)
        ok = True
        for bi, B in enumerate(plan["blocks"]):
            v = bench.frame_shard_rank_view(plan, B, rank)
            # a row = [sequence, absolute frame, producing rank] + payload derived from them: what a record carries
            rows = torch.tensor([[q_, B["t0"] + k, rank, (q_ * 1000 + B["t0"] + k) % 251] for q_, k in v["mine"]], dtype=torch.int32).view(-1, 4)
            send = rows[torch.tensor(v["send_perm"], dtype=torch.int64)] if len(v["mine"]) else rows
            got, work = bench.exchange_rows(dist, world, send, v["send_splits"], v["recv_splits"])
            if work is not None:
                work.wait()
            pool = got[torch.tensor(v["pool_index"], dtype=torch.int64)]
            S = plan["S"]
            for k in range(B["n"]):
                for l, q_ in enumerate(plan["lanes"][rank]):
                    row = pool[k * S + l].tolist()
                    ok = ok and row[0] == q_ and row[1] == B["t0"] + k and row[3] == (q_ * 1000 + B["t0"] + k) % 251
                    src = [r for r in range(world) if B["piece"][r][0] <= B["units"].index((q_, k)) < B["piece"][r][1]]
                    ok = ok and row[2] == src[0]
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_frame_shard_hand_over_gloo_world2():
    """The block hand-over (bench.exchange_rows: all_to_all_single with split sizes) between two real processes: every owner ends up with the rows of
    its sequences' frames, in pool order, whichever rank produced them."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_frame_shard_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    out = sorted(q.get(timeout=120) for _ in ps)
    for p in ps: p.join(60)
    assert out == [(0, True), (1, True)]
