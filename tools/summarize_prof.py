#!/usr/bin/env python3
"""Condense rocprofv3 output merged under gpurun_out/ into the small files committed under profiles/.

  python tools/summarize_prof.py <round-tag> <kernel-trace-dir> [<fetch-pmc-dir> <write-pmc-dir> <batch_images> [<workload> [<sq-pmc-dir>]]]

Writes profiles/<tag>_kernel_stats.csv (our kernels + copies, torch template names shortened) and, when the two
PMC passes are given, profiles/<tag>_pmc.csv plus profiles/pmc_traffic.json (read by bench.py):
hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — FETCH_SIZE/WRITE_SIZE are in KiB and gfx950's
FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section), so the read side is doubled.
"""
import collections
import csv
import glob
import json
import os
import sys

tag, kt = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
newest = lambda pattern_dir, pat: max(glob.glob(os.path.join(pattern_dir, "**", pat), recursive=True), key=os.path.getmtime)   # gpurun merges runs: take the last one
stats = newest(kt, "*_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(root, "profiles", tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        name = r["Name"].split("(")[0]
        if len(name) > 60:
            name = name[:57] + "..."
        w.writerow([name, r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]), r["Percentage"], r["MinNs"],
                    r["MaxNs"]])
print("wrote", tag + "_kernel_stats.csv")
if len(sys.argv) >= 6:
    agg = collections.defaultdict(list)
    for d in (sys.argv[3], sys.argv[4]):
        f = newest(d, "*_counter_collection.csv")
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("k_") or k.startswith("void k_"):
                agg[(k.replace("void ", "").split("<")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    kernels = sorted({k for k, _ in agg})
    traffic = {}
    with open(os.path.join(root, "profiles", tag + "_pmc.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Launches", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg",
                    "hbm_bytes_per_launch=(2*FETCH+WRITE)*1024"])
        for k in kernels:
            fe = agg.get((k, "FETCH_SIZE"), [0]); wr = agg.get((k, "WRITE_SIZE"), [0])
            fa, wa = sum(fe) / len(fe), sum(wr) / len(wr)
            hb = int((2 * fa + wa) * 1024)
            w.writerow([k, len(fe), "%.1f" % fa, "%.1f" % wa, hb])
            traffic[k] = {"hbm_bytes_per_launch": hb, "batch_images": int(sys.argv[5]), "profile": tag + "_pmc.csv"}
    workload = sys.argv[6] if len(sys.argv) >= 7 else "stereo-yolo"
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    try:
        allw = json.load(open(path))
        if allw and not all(isinstance(v, dict) and all(isinstance(x, dict) for x in v.values()) for v in allw.values()):
            allw = {}
        if any("hbm_bytes_per_launch" in v for v in allw.values()):          # the flat round-2 layout
            allw = {}
    except Exception:
        allw = {}
    allw[workload] = traffic                                               # bench.py reads the table of the workload it runs
    json.dump(allw, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", tag + "_pmc.csv, pmc_traffic.json[%s]" % workload)
    if len(sys.argv) >= 8:                                                 # SQ instruction counters (their own pass)
        f = newest(sys.argv[7], "*_counter_collection.csv")
        sq = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if k.startswith("k_"):
                sq[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(os.path.join(root, "profiles", tag + "_pmc_sq.csv"), "w", newline="") as fo:
            w = csv.writer(fo)
            names = sorted({c for v in sq.values() for c in v})
            w.writerow(["Kernel", "Launches"] + [n + "_avg" for n in names])
            for k in sorted(sq):
                n = max(len(v) for v in sq[k].values())
                w.writerow([k, n] + ["%.0f" % (sum(sq[k][c]) / max(1, len(sq[k][c]))) for c in names])
        print("wrote", tag + "_pmc_sq.csv")
