#!/bin/bash
# PMC passes over tools/pmc_pk.py (persistent GEMM vs round-1 kernels): SQ wait / LDS counters, then cache counters.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_pk
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python $R/tools/pmc_pk.py > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/p2 -o p2 --output-format csv -- python $R/tools/pmc_pk.py > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS --kernel-trace -d $OUT/p3 -o p3 --output-format csv -- python $R/tools/pmc_pk.py > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "gemm" not in kn:
            continue
        short = kn.split("(")[0].replace("void (anonymous namespace)::", "")[:60]
        key = (short, r["Grid_Size"], r.get("LDS_Block_Size", ""))
        rows.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$R/gpurun_out/pmc_pk_summary.txt", "w") as out:
    for k, d in rows.items():
        out.write(str(k) + "\n")
        for c in sorted(d):
            v = d[c]
            out.write(f"    {c:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}\n")
print(open("$R/gpurun_out/pmc_pk_summary.txt").read()[:200])
PY
