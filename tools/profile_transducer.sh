#!/bin/bash
# kernel trace + gap analysis of tools/bench_transducer.py (config 4), run on the GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_rnnt
mkdir -p $OUT
python $R/tools/bench_transducer.py --steps 12 2>/dev/null | tail -1 | cut -c1-400
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o rnnt -- python $R/tools/bench_transducer.py --steps 8 > $OUT.log 2>&1
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/prof_rnnt_summary.txt > /dev/null
python $R/tools/gap_analysis.py $DB 6 > $R/gpurun_out/prof_rnnt_gaps.txt 2>&1
head -28 $R/gpurun_out/prof_rnnt_summary.txt | cut -c1-190
head -12 $R/gpurun_out/prof_rnnt_gaps.txt
