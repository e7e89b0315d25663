#!/bin/bash
# PMC passes over tools/prof_flash.py (fused attention kernels at the headline shape); per-kernel averages -> gpurun_out/pmc_flash_summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_flash
rm -rf $OUT; mkdir -p $OUT
run() { timeout 300 rocprofv3 --pmc "${@:2}" --kernel-trace -d $OUT/$1 -o $1 --output-format csv -- python $R/tools/prof_flash.py > $OUT/$1.log 2>&1; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16
run p3 SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY
python - <<PY > $R/gpurun_out/pmc_flash_summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "flash" not in k and "rp_" not in k and "keep_bits" not in k:
            continue
        k = k.replace("void (anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:34s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
cat $R/gpurun_out/pmc_flash_summary.txt
rm -rf $OUT
