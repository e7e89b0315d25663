"""GPU idle-gap analysis of a rocprofv3 kernel trace (rocpd sqlite): busy time vs wall per training step (steps are
delimited by the adam kernel), histogram of the idle gaps and the largest ones.  Usage: gap_analysis.py <db> [nsteps]"""
import sqlite3, sys
import numpy as np

db = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
c = sqlite3.connect(db)
rows = c.execute("select start, end, name from kernels order by start").fetchall()
adam = [r for r in rows if "adam_kernel" in r[2]]
if len(adam) > nsteps:
    t0, t1 = adam[-nsteps - 1][1], adam[-1][1]
else:  # not a training trace (e.g. the decode block): the whole trace after the first tenth (start-up) as ONE "step"
    nsteps = 1
    t0, t1 = rows[len(rows) // 10][0], rows[-1][1]
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
busy, cur_end, gaps, prev = 0, t0, [], "<step start>"
for s, e, n in sel:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
    prev = n
print(f"per step: wall {(t1 - t0) / nsteps / 1e6:.2f} ms, GPU busy (union) {busy / nsteps / 1e6:.2f} ms, "
      f"sum of kernel durations {sum(e - s for s, e, _ in sel) / nsteps / 1e6:.2f} ms, kernels {len(sel) / nsteps:.0f}")
g = np.array([x[0] for x in gaps]) / 1e3
print(f"idle gaps: {len(gaps) / nsteps:.0f} per step, {g.sum() / nsteps / 1e3:.2f} ms per step")
for lo, hi in [(0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 1000), (1000, 1e9)]:
    m = (g >= lo) & (g < hi)
    print(f"  gaps {lo}-{hi} us: {m.sum() / nsteps:.0f} per step, {g[m].sum() / nsteps / 1e3:.2f} ms per step")
for d, p, n in sorted(gaps, key=lambda x: -x[0])[:20]:
    print(f"{d / 1e3:9.1f} us  after {p[:70]:70s} before {n[:60]}")

# kernels that run concurrently with the optimizer kernel (what shares HBM with the 2.7 GB Adam pass)
ov = {}
for a_s, a_e, _ in adam[-nsteps:]:
    for s, e, n in rows:
        if "adam_kernel" in n:
            continue
        o = min(e, a_e) - max(s, a_s)
        if o > 0:
            ov[n] = ov.get(n, 0) + o
if ov:
    print("kernels overlapping adam_kernel (us per step):")
    for n, o in sorted(ov.items(), key=lambda kv: -kv[1])[:8]:
        print(f"{o / nsteps / 1e3:9.1f} us  {n[:110]}")
else:
    print("no kernel overlaps adam_kernel")
