#!/bin/bash
# PMC passes over tools/pmc_gemm.py; summaries land in gpurun_out/pmc_gemm/*.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python $R/tools/pmc_gemm.py 3 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "gemm_bf16" not in name and "splitk" not in name:
            continue
        key = (name[:60], r.get("Grid_Size"), r.get("LDS_Block_Size"))
        d = rows[key]
        c = r["Counter_Name"]
        d.setdefault(c, []).append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for k, d in rows.items():
        print(k, file=o)
        for c, v in sorted(d.items()):
            print(f"    {c:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}", file=o)
print(open("$OUT/summary.txt").read())
PY
