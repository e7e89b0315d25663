"""Aggregate a `bench.py --gemm-dump` file per GEMM shape: launches, total ms, average µs and TFLOP/s, sorted by time."""
import sys
from collections import defaultdict

rows = defaultdict(lambda: [0, 0.0])
for line in open(sys.argv[1]):
    M, N, K, b, aks, bks, sk, bm64, epi, ms = line.split()
    key = (int(M), int(N), int(K), int(b), int(aks), int(bks), int(sk), int(bm64), int(epi))
    rows[key][0] += 1
    rows[key][1] += float(ms)
# epilogue bit 128 = a grouped weight-gradient launch: its record carries (workgroups, problems, reduction length), not a GEMM
# shape — its flops are only known to the library (bench.py's roofline uses them); it is listed by time and left out of TFLOP/s
tot = sum(v[1] for v in rows.values())
plain = {k: v for k, v in rows.items() if not (k[8] & 128)}
ptot = sum(v[1] for v in plain.values())
totfl = sum(2.0 * k[0] * k[1] * k[2] * k[3] * v[0] for k, v in plain.items())
print(f"# {sum(v[0] for v in rows.values())} launches, {tot:.3f} ms; plain GEMM launches: {ptot:.3f} ms, {totfl / ptot / 1e9:.1f} TFLOP/s")
print(f"# {'M':>7} {'N':>6} {'K':>7} {'batch':>5} ks sk bm64 epi {'n':>5} {'tot_ms':>8} {'pct':>5} {'avg_us':>8} {'TF/s':>7}")
for k, (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    fl = 0.0 if (k[8] & 128) else 2.0 * k[0] * k[1] * k[2] * k[3]
    print(f"  {k[0]:7d} {k[1]:6d} {k[2]:7d} {k[3]:5d} {k[4]}{k[5]} {k[6]:2d} {k[7]:4d} {k[8]:3d} {n:5d} {ms:8.3f} {100 * ms / tot:5.1f} {1e3 * ms / n:8.1f} {fl * n / ms / 1e9:7.1f}")
