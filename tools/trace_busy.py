#!/usr/bin/env python3
"""Developer tool: from a rocprofv3 kernel trace (csv), how busy the detector's convolution stream was.
usage: trace_busy.py <kernel_trace.csv> [name-substring=k_conv]  -> span, summed kernel time, the largest gaps between consecutive matching kernels."""
import csv, sys
f, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_conv")
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
sel = [r for r in rows if pat in r[2]]
t0, t1 = sel[0][0], sel[-1][1]
busy = sum(e - s for s, e, _ in sel)
print("kernels matching %r: %d, span %.1f ms, summed %.1f ms (%.1f %%)" % (pat, len(sel), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
gaps = sorted(((sel[i + 1][0] - sel[i][1]) / 1e6, (sel[i][1] - t0) / 1e6) for i in range(len(sel) - 1))
big = [g for g in gaps if g[0] > 0.3]
print("gaps > 0.3 ms between consecutive matching kernels: %d, total %.1f ms; the 12 largest (ms, at ms):" % (len(big), sum(g[0] for g in big)))
print([(round(a, 2), round(b, 1)) for a, b in gaps[-12:]])
other = {}
for s, e, n in rows:
    if pat in n or s < t0 or e > t1: continue
    k = n.split("(")[0][:40]
    other[k] = other.get(k, 0) + (e - s)
print("other kernels inside the span (ms):", sorted(((round(v / 1e6, 1), k) for k, v in other.items()), reverse=True)[:14])
