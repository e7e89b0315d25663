#!/bin/bash
# usage: bash tools/profile_cmd.sh <tag> <steps-for-gap-analysis> <python script + args...>   (run on the GPU box)
TAG=$1; NS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python $R/$@ 2>/dev/null | tail -1 | cut -c1-420
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $R/$@ > $OUT.log 2>&1
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/${TAG}_summary.txt > /dev/null
python $R/tools/gap_analysis.py $DB $NS > $R/gpurun_out/${TAG}_gaps.txt 2>&1
head -30 $R/gpurun_out/${TAG}_summary.txt | cut -c1-180
head -3 $R/gpurun_out/${TAG}_gaps.txt
