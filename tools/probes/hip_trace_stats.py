"""Steady-state host cost per HIP runtime call from a rocprofv3 --hip-runtime-trace CSV: the last `frac` of the trace by time."""
import csv, sys, collections, statistics
path, frac = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = []
with open(path) as f:
    r = csv.DictReader(f)
    for x in r:
        rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Function"], x.get("Thread_Id", "0")))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - frac * (t1 - t0)
sel = [x for x in rows if x[0] >= cut]
span_ms = (sel[-1][1] - sel[0][0]) / 1e6
by = collections.defaultdict(list)
for s, e, fn, tid in sel:
    by[fn].append((e - s) / 1e3)
print(f"window {span_ms:.1f} ms, {len(sel)} calls; threads: {collections.Counter(x[3] for x in sel).most_common(4)}")
print(f"{'function':44s} {'calls':>7s} {'total ms':>9s} {'% window':>8s} {'median us':>9s} {'mean us':>8s} {'p90 us':>8s}")
for fn, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:16]:
    v.sort()
    print(f"{fn:44s} {len(v):7d} {sum(v)/1e3:9.2f} {100*sum(v)/1e3/span_ms:8.1f} {statistics.median(v):9.2f} {sum(v)/len(v):8.2f} {v[int(0.9*len(v))]:8.2f}")

# --- where the long launches are -------------------------------------------------------------------------------------------
L = [(s, e, tid) for s, e, fn, tid in sel if fn == "hipLaunchKernel"]
bins = [(0, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 1000), (1000, 1e9)]
print("hipLaunchKernel duration histogram (us): calls, total ms")
for lo, hi in bins:
    v = [(e - s) / 1e3 for s, e, t in L if lo <= (e - s) / 1e3 < hi]
    print(f"  {lo:>5} - {hi:<6g}: {len(v):6d} {sum(v)/1e3:8.2f}")
# last quarter of the window only (certainly steady state)
cut2 = sel[-1][1] - 0.25 * (sel[-1][1] - sel[0][0])
L2 = [(s, e, t) for s, e, t in L if s >= cut2]
tot = sum(e - s for s, e, t in L2) / 1e6
print(f"last quarter: {len(L2)} launches, {tot:.2f} ms in hipLaunchKernel of {(sel[-1][1]-cut2)/1e6:.1f} ms; per thread: "
      + str({t: round(sum(e - s for s, e, tt in L2 if tt == t) / 1e6, 2) for t in set(x[2] for x in L2)}))
long_ = sorted(L2, key=lambda x: x[0] - x[1])[:25]
print("longest launches of the last quarter: start offset ms, duration us, thread")
for s, e, t in sorted(long_):
    print(f"  {(s - cut2)/1e6:9.3f} {(e - s)/1e3:9.1f} {t}")

# --- host time BETWEEN runtime calls (the caller's own code), last quarter, per thread ---------------------------------------
per_thread = collections.defaultdict(list)
for s, e, fn, tid in sel:
    if s >= cut2:
        per_thread[tid].append((s, e, fn))
for tid, v in per_thread.items():
    v.sort()
    inside = sum(e - s for s, e, fn in v) / 1e6
    gaps = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
    small = [g for g in gaps if g < 50]
    print(f"thread {tid}: {len(v)} calls, {inside:.1f} ms inside the runtime, gaps < 50 us: {len(small)} totalling {sum(small)/1e3:.1f} ms "
          f"(median {statistics.median(small):.2f} us), gaps >= 50 us: {len(gaps)-len(small)} totalling {sum(g for g in gaps if g >= 50)/1e3:.1f} ms")
    launches = [i for i in range(len(v) - 1) if v[i][2] == "hipLaunchKernel" and v[i + 1][2] in ("hipLaunchKernel", "hipGetLastError")]
    # gap from one launch's end to the next LAUNCH's start (skipping the cheap bookkeeping calls in between)
    idx = [i for i in range(len(v)) if v[i][2] == "hipLaunchKernel"]
    l2l = [(v[idx[k + 1]][0] - v[idx[k]][1]) / 1e3 for k in range(len(idx) - 1)]
    l2l_small = sorted(g for g in l2l if g < 50)
    if l2l_small:
        print(f"   launch-to-launch host time (< 50 us): n {len(l2l_small)}, median {statistics.median(l2l_small):.2f} us, mean {sum(l2l_small)/len(l2l_small):.2f} us, "
              f"p90 {l2l_small[int(0.9*len(l2l_small))]:.2f} us, total {sum(l2l_small)/1e3:.1f} ms")
