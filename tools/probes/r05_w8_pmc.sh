#!/bin/bash
# round 5: PMC passes over single cases of tools/probes/gemm_w8_timing (arguments: case indices); summary -> gpurun_out/r05/w8_pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-v1}
OUT=$R/gpurun_out/r05/pmc_$TAG
mkdir -p $OUT
for CASE in "$@"; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --kernel-trace -d $OUT/c${CASE}_p$i -o p --output-format csv -- $R/tools/probes/gemm_w8_timing 6240 $CASE > $OUT/c${CASE}_p$i.log 2>&1
    echo "case $CASE pass $i rc=$?"
  done
done
python3 - <<PY
import csv, glob, collections, re
out = open("$R/gpurun_out/r05/w8_pmc_$TAG.txt", "w")
for case in "$*".split():
    d = collections.defaultdict(list)
    for f in sorted(glob.glob("$OUT/c%s_p*/**/*counter_collection.csv" % case, recursive=True)):
        for r in csv.DictReader(open(f)):
            if "gemm_w8" not in r["Kernel_Name"]:
                continue
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    log = open(glob.glob("$OUT/c%s_p1.log" % case)[0]).read().strip().splitlines()
    print("case", case, "|", (log[-1][:60] if log else ""), file=out)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}", file=out)
out.close()
print(open("$R/gpurun_out/r05/w8_pmc_$TAG.txt").read())
PY
