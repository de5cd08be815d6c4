"""Per-dispatch table of the detector's kernels from a rocprofv3 results db (developer tool).
usage: yolo_layer_table.py <results.db> <batch>"""
import sqlite3, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
db, B = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
last = [i for i, r in enumerate(rows) if 'k_blob' in r[0]][-1]
# the convolution list of yolov3.cfg at 640x480: (layer, size, stride, cin, filters, Ho, Wo)
convs = []
state = {"i": 0, "h": 480, "w": 640, "c": 3}
def conv(f, k, s):
    st = state
    ho, wo = (st["h"] + s - 1) // s, (st["w"] + s - 1) // s
    convs.append((st["i"], k, s, st["c"], f, ho, wo)); st["h"], st["w"], st["c"] = ho, wo, f; st["i"] += 1
def skip(n=1): state["i"] += n
def res(c, n):
    for _ in range(n): conv(c // 2, 1, 1); conv(c, 3, 1); skip()
conv(32, 3, 1); conv(64, 3, 2); res(64, 1); conv(128, 3, 2); res(128, 2); conv(256, 3, 2); res(256, 8)
conv(512, 3, 2); res(512, 8); conv(1024, 3, 2); res(1024, 4)
for _ in range(3): conv(512, 1, 1); conv(1024, 3, 1)
conv(255, 1, 1); skip()
skip(); state["c"] = 512; conv(256, 1, 1); skip(); state.update(h=30, w=40); skip(); state["c"] = 768
for _ in range(3): conv(256, 1, 1); conv(512, 3, 1)
conv(255, 1, 1); skip()
skip(); state["c"] = 256; conv(128, 1, 1); skip(); state.update(h=60, w=80); skip(); state["c"] = 384
for _ in range(3): conv(128, 1, 1); conv(256, 3, 1)
conv(255, 1, 1); skip()
ci = 0
tot = 0
pre = 0.0       # the Winograd input transform that precedes its GEMM (mode f32w): counted into the layer
for r in rows[last:]:
    name = r[0]; us = (r[2] - r[1]) / 1e3
    if 'k_wino_input' in name:
        pre = us
    elif ('conv' in name or 'k_wino_gemm' in name) and ci < len(convs):
        i, k, s, cin, f, ho, wo = convs[ci]; ci += 1
        fl = 2.0 * k * k * cin * f * ho * wo * B
        wino = 'k_wino_gemm' in name
        ex = 2.0 * 16 * cin * f * ((ho + 1) // 2) * ((wo + 1) // 2) * B if wino else fl
        tot += us + pre
        print("L%3d %dx%d/%d %4d->%4d %3dx%3d  %-22s grid %5d x %2d  %8.1f us  %6.1f TFLOP/s nominal%s" % (i, k, k, s, cin, f, wo, ho, name[:22], r[3] // r[5], r[4], us + pre, fl / (us + pre) / 1e6,
              "  (input transform %.1f us; GEMM alone %.1f TFLOP/s executed)" % (pre, ex / us / 1e6) if wino else ""))
        pre = 0.0
    else:
        print("     %-40s %8.1f us" % (name[:40], us))
print("conv total %.2f ms" % (tot / 1e3))
