# kernel-trace the bench in two trees on the same box and print the per-kernel totals side by side
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in . _prev; do
  O=$R/gpurun_out/cmp_$(echo $d | tr -d './_')x
  rm -rf $O; mkdir -p $O
  (cd $R/$d && timeout 600 rocprofv3 --kernel-trace -d $O -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-other-configs --no-roofline > $O.log 2>&1)
  DB=$(ls $O/*.db $O/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB $O.txt > /dev/null
  tail -1 $O.log | cut -c1-120
done
python - <<PY
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s*([\d.]+)\s+([\d.]+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", l)
        if m: d[m.group(7)[:70]]=(float(m.group(1)), int(m.group(3)))
    return d
a=load("$R/gpurun_out/cmp_x.txt"); b=load("$R/gpurun_out/cmp_prevx.txt")
rows=[]
for k in set(a)|set(b):
    ta,ca=a.get(k,(0,0)); tb,cb=b.get(k,(0,0))
    rows.append((ta-tb,k,ta,ca,tb,cb))
rows.sort(key=lambda r:-abs(r[0]))
print("diff_ms(new-prev)  new_ms calls | prev_ms calls   kernel   [13 steps]")
for r in rows[:25]: print(f"{r[0]:8.2f}  {r[2]:8.2f} {r[3]:5d} | {r[4]:8.2f} {r[5]:5d}  {r[1][:60]}")
print("total", sum(v[0] for v in a.values()), sum(v[0] for v in b.values()))
PY
