"""Per-kernel PMC counter averages from a rocprofv3 --pmc run (rocpd sqlite).  Usage: pmc_summary.py db [substr]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
print(cols)
q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name"
try:
    for r in cur.execute(q, (f"%{sub}%",)):
        print(f"{r[0][:60]:60s} {r[1]:32s} {r[2]:16.1f} n={r[3]}")
except Exception as e:
    print("ERR", e)
    for r in cur.execute("select * from counters_collection limit 3"): print(r)
