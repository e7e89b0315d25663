#!/bin/bash
# rocprofv3 kernel trace of the default bench command + summaries (gpurun_out/profN_*): run on the GPU box.
#   bash tools/profile_bench.sh <tag> [steps]
TAG=${1:-prof}
STEPS=${2:-12}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-decode --no-other-configs --no-roofline > $OUT.log 2>&1
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
echo "db=$DB"
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/${TAG}_summary.txt > /dev/null
python $R/tools/gap_analysis.py $DB 8 > $R/gpurun_out/${TAG}_gaps.txt 2>&1
python $R/tools/stream_analysis.py $DB 8 > $R/gpurun_out/${TAG}_streams.txt 2>&1
tail -1 $OUT.log | cut -c1-300
head -45 $R/gpurun_out/${TAG}_summary.txt | cut -c1-200
head -12 $R/gpurun_out/${TAG}_gaps.txt
