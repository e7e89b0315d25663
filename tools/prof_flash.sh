#!/bin/bash
# per-kernel time of the fused attention kernels at the headline shape: bash tools/prof_flash.sh [B] [T]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_flash
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o fl -- python $R/tools/bench_flash.py ${1:-21} ${2:-308} > $OUT.log 2>&1
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/prof_flash_summary.txt > /dev/null
grep -E "flash|rp_|keep_bits" $R/gpurun_out/prof_flash_summary.txt | cut -c1-150
rm -rf $OUT
