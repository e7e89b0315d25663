"""In-step view of the forward / data-gradient GEMM launches from a rocprofv3 kernel trace of bench.py (rocpd sqlite): per shape
class, launches per step and the average duration ALONE on the device vs NEXT TO a kernel of another queue (the layer runtime's
side stream), in the forward and in the backward half of the step.  Together with tools/bench_gemm_shapes.py (isolated hot / cold,
vendor ceiling) this is profiles/r04_gemm_shapes.txt.

A launch is classified by kernel, grid and position: grid.x = N / 128 column tiles, grid.y = row blocks of 64; the reduction length
follows from the kernel the dispatcher picks and the half of the step (forward: before the CTC scan; backward: after) — the table
below names the layer operation each class is.   Usage: gemm_instep_report.py <db> [nsteps]"""
import re
import sqlite3
import sys

db = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next(k for k in ("queue_id", "stream_id") if k in cols)
rows = c.execute(f"select start, end, name, {qcol}, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
adam = [r for r in rows if "adam_kernel" in r[2]]
t0, t1 = adam[-nsteps - 1][1], adam[-1][1]
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
busy = {}
for r in sel:
    busy[r[3]] = busy.get(r[3], 0) + r[1] - r[0]
main = max(busy, key=busy.get)
others = sorted((r[0], r[1]) for r in sel if r[3] != main)
scan = sorted(r[0] for r in sel if "ctc_scan_kernel" in r[2])


def other_busy(a, b):
    t, cur = 0, a
    for s, e in others:
        if e <= cur:
            continue
        if s >= b:
            break
        lo, hi = max(s, cur), min(e, b)
        if hi > lo:
            t += hi - lo
            cur = hi
    return t


def phase(ts):  # forward: between a step's start and its CTC scan
    import bisect

    i = bisect.bisect_right(scan, ts)
    prev_adam = max((a[1] for a in adam if a[1] <= ts), default=t0)
    return "fwd" if i < len(scan) and (i == 0 or scan[i - 1] < prev_adam) else "bwd"


NAMES = {
    ("reg", 16, "fwd"): "6.2k x 2048 x 512  ffn W1 fwd (bias, SiLU, dropout, 2 outputs)",
    ("reg", 16, "bwd"): "6.2k x 2048 x 512  ffn W2 dgrad (dropout * SiLU'(aux))",
    ("reg", 12, "fwd"): "6.2k x 1536 x 512  qkv projection (bias, query split)",
    ("reg", 8, "fwd"): "6.2k x 1024 x 512  conv pointwise 1",
    ("glds", 4, "fwd"): "6.2k x 512 x {512,2048,2560}  W2 fwd / out_proj / pointwise 2 / fc0 / pos_proj",
    ("glds", 4, "bwd"): "6.2k x 512 x {512..2048}  dgrads with a 512-wide output (W1, qkv, out_proj, pw1, pw2)",
    ("reg", 40, "fwd"): "6.2k x 5004 x 512  fc_out",
    ("reg", 20, "bwd"): "6.2k x 2560 x 512  fc0 dgrad",
}
stats = {}
for s, e, n, q, gx, gy, gz, wx in sel:
    if q != main or not ("gemm_bf16_kernel" in n or "gemm_glds_kernel" in n):
        continue
    kind = "glds" if "glds" in n else "reg"
    m = re.search(r"<([^>]*)>", n)
    if gy < 80:  # (positional-table projections and the shortest batches: not the recipe's full 26 000-frame batches)
        continue
    key = (kind, m.group(1) if m else "", gx // max(1, wx), phase(s))
    st = stats.setdefault(key, [0, 0.0, 0, 0.0, 0])
    st[4] += gy
    if other_busy(s, e) > 0.5 * (e - s):
        st[2] += 1
        st[3] += e - s
    else:
        st[0] += 1
        st[1] += e - s
print(f"# main-queue GEMM launches over {nsteps} steps of {db.split('/')[-2] if '/' in db else db}; us per launch")
print("# (launches with >= 80 row blocks of 64, i.e. the full batches; rowblk = their mean number of row blocks)")
print(f"# {'kernel':5s} {'template':16s} {'ntile':>5s} {'rowblk':>6s} {'half':4s} {'per step':>8s} {'alone us':>9s} {'(n)':>6s} {'co-run us':>9s} {'(n)':>6s}  operation")
tot = 0.0
for key, st in sorted(stats.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    kind, tpl, nt, ph = key
    n = st[0] + st[2]
    rb = round(st[4] / max(1, n))
    tot += st[1] + st[3]
    name = NAMES.get((kind, nt, ph), "")
    print(f"  {kind:5s} {tpl:16s} {nt:5d} {rb:6d} {ph:4s} {n / nsteps:8.1f} {st[1] / max(1, st[0]) / 1e3:9.1f} {st[0]:6d} "
          f"{st[3] / max(1, st[2]) / 1e3:9.1f} {st[2]:6d}  {name}")
print(f"# total {tot / nsteps / 1e6:.2f} ms per step")
