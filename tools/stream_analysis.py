"""Per-queue view of a rocprofv3 kernel trace (rocpd sqlite) of the training step: for each HIP stream (hardware queue) the busy
time per step, and for the busiest queue (the main stream) its idle gaps split by whether another queue was busy during the gap
(= the main stream waited for side-stream work or for a cross-stream event) or the whole device was idle (= host / launch latency).
Steps are delimited by the adam kernel.   Usage: stream_analysis.py <db> [nsteps]"""
import sqlite3
import sys

db = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((k for k in ("queue_id", "stream_id", "queue", "stream") if k in cols), None)
print("kernels view columns:", cols)
if qcol is None:
    sys.exit("no queue / stream column in this trace")
rows = c.execute(f"select start, end, name, {qcol} from kernels order by start").fetchall()
adam = [r for r in rows if "adam_kernel" in r[2]]
t0, t1 = adam[-nsteps - 1][1], adam[-1][1]
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
queues = {}
for s, e, n, q in sel:
    queues.setdefault(q, []).append((s, e, n))
print(f"per step over {nsteps} steps: wall {(t1 - t0) / nsteps / 1e6:.2f} ms")
order = sorted(queues, key=lambda q: -sum(e - s for s, e, _ in queues[q]))
for q in order:
    ks = queues[q]
    busy = sum(e - s for s, e, _ in ks)
    print(f"  queue {q}: {len(ks) / nsteps:6.0f} kernels per step, busy {busy / nsteps / 1e6:6.2f} ms per step")
main = order[0]
others = sorted((s, e) for q in order[1:] for s, e, _ in queues[q])


def other_busy(a, b):
    """time in [a, b) during which some other queue runs a kernel"""
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


mk = sorted(queues[main])
wait_other = wait_idle = 0
big = []
for (s0, e0, n0), (s1, e1, n1) in zip(mk, mk[1:]):
    gap = s1 - e0
    if gap <= 0:
        continue
    ob = other_busy(e0, s1)
    wait_other += ob
    wait_idle += gap - ob
    big.append((gap, ob, n0, n1))
print(f"main queue {main}: gaps while another queue is busy {wait_other / nsteps / 1e6:.2f} ms per step, "
      f"gaps with the device idle {wait_idle / nsteps / 1e6:.2f} ms per step")
# main-queue kernel time with / without a co-running kernel of another queue
co = 0
for s, e, _ in mk:
    co += other_busy(s, e)
tot = sum(e - s for s, e, _ in mk)
print(f"main queue kernel time: {tot / nsteps / 1e6:.2f} ms per step, of which {co / nsteps / 1e6:.2f} ms with another queue's kernel running")
print("largest main-queue gaps (us, of which another queue busy):")
for gap, ob, n0, n1 in sorted(big, key=lambda x: -x[0])[:25]:
    print(f"{gap / 1e3:8.1f} {ob / 1e3:8.1f}  after {n0[:60]:60s} before {n1[:60]}")
# main-queue kernels: average duration alone vs co-running
stats = {}
for s, e, n in mk:
    ob = other_busy(s, e)
    key = n[:70]
    st = stats.setdefault(key, [0, 0.0, 0, 0.0])
    if ob > 0.5 * (e - s):
        st[2] += 1
        st[3] += e - s
    else:
        st[0] += 1
        st[1] += e - s
print("main-queue kernels, average us alone / with a co-running kernel of another queue (calls):")
for key, (n0, d0, n1, d1) in sorted(stats.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:22]:
    a = d0 / n0 / 1e3 if n0 else 0.0
    b = d1 / n1 / 1e3 if n1 else 0.0
    print(f"  {a:8.1f} ({n0:5d})  {b:8.1f} ({n1:5d})  {key}")
