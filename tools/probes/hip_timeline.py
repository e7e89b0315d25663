"""One stretch of the main thread's HIP runtime calls with the kernel each launch dispatched: start offset, duration, gap to the next call.
usage: hip_timeline.py <hip_api_trace.csv> <kernel_trace.csv> [n_calls] [skip_fraction]"""
import csv, sys, collections
api, ker = sys.argv[1], sys.argv[2]
ncalls = int(sys.argv[3]) if len(sys.argv) > 3 else 200
skip = float(sys.argv[4]) if len(sys.argv) > 4 else 0.8
kn = {}
with open(ker) as f:
    for x in csv.DictReader(f):
        kn[x["Correlation_Id"]] = x["Kernel_Name"]
rows = []
with open(api) as f:
    for x in csv.DictReader(f):
        rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Function"], x["Thread_Id"], x["Correlation_Id"]))
rows.sort()
main = collections.Counter(r[3] for r in rows).most_common(2)
t0, t1 = rows[0][0], rows[-1][1]
cut = t0 + skip * (t1 - t0)
cheap = {"hipGetDevice", "hipSetDevice", "hipGetLastError", "hipEventQuery", "__hipPushCallConfiguration", "__hipPopCallConfiguration"}
for tid, _ in main:
    v = [r for r in rows if r[3] == tid and r[0] >= cut and r[2] not in cheap][:ncalls]
    if not v:
        continue
    print(f"--- thread {tid}")
    for i, (s, e, fn, _, cid) in enumerate(v):
        gap = (v[i + 1][0] - e) / 1e3 if i + 1 < len(v) else 0.0
        name = kn.get(cid, "")
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        print(f"{(s - v[0][0]) / 1e3:9.1f} us  {fn:22s} {(e - s) / 1e3:6.1f} us  gap {gap:6.1f}  {name}")
