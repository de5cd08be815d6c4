#!/usr/bin/env python3
"""Condense the detector's rocprofv3 runs (tools/bench_yolo.py under --kernel-trace, and three --pmc passes) into
profiles/<tag>_yolo_pmc.csv.

  python tools/summarize_yolo_prof.py <tag> <kernel-trace-dir> <mfma-pmc-dir> <fetch-pmc-dir> <write-pmc-dir>

MFMA FLOP/s per kernel = SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 / kernel time (the counter ticks once per 512 FLOPs: checked
against the layers' arithmetic, 510 FLOPs per tick on k_conv3x3_glds); utilisation = that / 2.5 PFLOP/s dense f16.
HBM bytes per call = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B).
"""
import collections
import csv
import glob
import os
import sys

tag, kt, pm, pf, pw = sys.argv[1:6]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import re


def clean(name):
    name = name.split("(")[0].replace("void ", "")
    m_ = re.match(r"_Z(\d+)", name)                  # rocprofv3 leaves some names mangled: _Z<len><name>...
    if m_:
        n_ = int(m_.group(1)); st = m_.end()
        return name[st:st + n_]
    return name


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        agg[clean(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg


m, f, w = load(pm), load(pf), load(pw)
dur = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(os.path.join(kt, "**", "*_kernel_trace.csv"), recursive=True)[0])):
    dur[clean(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = os.path.join(root, "profiles", tag + "_yolo_pmc.csv")
with open(out, "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["Kernel", "Calls", "TotalMs", "AvgUs", "MfmaTFLOPs", "MfmaUtilPctOf2500", "FetchMiBPerCall_x2", "WriteMiBPerCall", "HbmGBps"])
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        if not k.startswith("k_"):
            continue
        n, tot = len(dur[k]), sum(dur[k])
        fl = m.get(k, {}).get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) * 512
        fe = 2 * f.get(k, {}).get("FETCH_SIZE", 0.0) / n / 1024
        we = w.get(k, {}).get("WRITE_SIZE", 0.0) / n / 1024
        tf = fl / (tot * 1e-6) / 1e12
        wr.writerow([k, n, "%.3f" % (tot / 1e3), "%.1f" % (tot / n), "%.1f" % tf, "%.1f" % (tf / 25.0), "%.1f" % fe, "%.1f" % we,
                     "%.0f" % ((fe + we) * 1.048576e6 / (tot / n * 1e-6) / 1e9)])
print(open(out).read())
