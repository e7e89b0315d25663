#!/bin/bash
# kernel durations of LayerNorm forward / backward at the headline shape under rocprofv3 (a dependent chain like in the layer:
# duration, not back-to-back throughput, is what the step pays): bash tools/prof_ln.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ln
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o ln -- python $R/tools/bench_misc.py > $OUT.log 2>&1
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/prof_ln_summary.txt > /dev/null
grep -E "ln_" $R/gpurun_out/prof_ln_summary.txt | cut -c1-130
rm -rf $OUT
