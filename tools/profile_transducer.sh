#!/bin/bash
# rocprofv3 kernel trace of the config-4 (Conformer-16 transducer) training step: bash tools/profile_transducer.sh <tag> [steps]
TAG=${1:-prof_td}
STEPS=${2:-10}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o td -- python $R/tools/bench_transducer.py --steps $STEPS --warmup 3 > $OUT.log 2>&1
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/${TAG}_summary.txt > /dev/null
python $R/tools/gap_analysis.py $DB 8 > $R/gpurun_out/${TAG}_gaps.txt 2>&1
rm -rf $OUT
tail -1 $OUT.log | cut -c1-300
head -50 $R/gpurun_out/${TAG}_summary.txt | cut -c1-200
head -12 $R/gpurun_out/${TAG}_gaps.txt
