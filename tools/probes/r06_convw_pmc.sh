#!/bin/bash
# round 6: L2 behaviour of conv_wgrad_kernel / conv_gather_kernel at the recipe shapes (TCC hit / miss / request counters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/convw_pmc
mkdir -p $O
for P in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE"; do
  i=$((i+1))
  EA_CONV_WGRAD_DEPTH=1 timeout 300 rocprofv3 --pmc $P --kernel-trace -d $O/p$i -o p --output-format csv -- python $R/tools/probes/r06_conv_wgrad_time.py > $O/p$i.log 2>&1
  echo "pass $i ($P) rc=$?"
done
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv_wgrad_kernel" in n:
            rows[(r["Grid_Size"] if "Grid_Size" in r else "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    print("grid", k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
