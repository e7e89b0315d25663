"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / avg / min / max.
Usage: python tools/rocpd_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    span = list(cur.execute("select min(start), max(end) from kernels"))[0]
    lines = [f"# rocprofv3 --kernel-trace summary of {sys.argv[1]}",
             f"# total kernel time {total / 1e6:.3f} ms over {len(rows)} kernels; first-to-last dispatch span {(span[1] - span[0]) / 1e6:.3f} ms",
             f"{'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>10} {'min_us':>9} {'max_us':>10}  name"]
    for name, n, tot, avg, mn, mx in rows:
        lines.append(f"{tot / 1e6:10.3f} {100.0 * tot / total:6.2f} {n:7d} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:10.2f}  {name[:140]}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out[:6000])


if __name__ == "__main__":
    main()
