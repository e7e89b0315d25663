#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gemm2
mkdir -p $OUT
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python $R/tools/pmc_gemm2.py > $OUT/p1.log 2>&1
python - <<PY
import csv, glob
rows = {}
for f in sorted(glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_glds" not in r["Kernel_Name"]:
            continue
        rows.setdefault((int(r["Dispatch_Id"]), r["Grid_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows):
    d = rows[k]
    print(k, {c: int(v) for c, v in d.items()}, "hit%%=%.1f" % (100 * d["TCC_HIT_sum"] / max(1, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])))
PY
