#!/bin/bash
# round 6: HIP API trace (host side) of the config-4 update step: which runtime calls the 24.5 ms of enqueue time are made of
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06/hiptrace_td
mkdir -p $OUT
timeout 900 rocprofv3 --hip-runtime-trace --stats -d $OUT -o td -f csv -- python $R/tools/bench_transducer.py --steps 12 --warmup 3 > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-300
ls -R $OUT | head -20
F=$(ls $OUT/*hip_api_stats.csv $OUT/*/*hip_api_stats.csv 2>/dev/null | head -1)
echo "stats file: $F"
head -40 "$F" | cut -c1-200 | tee $R/gpurun_out/r06/hiptrace_td_stats.txt
T=$(ls $OUT/*hip_api_trace.csv $OUT/*/*hip_api_trace.csv 2>/dev/null | head -1)
head -2 "$T" | cut -c1-400
python $R/tools/probes/hip_trace_stats.py "$T" 0.4 | tee $R/gpurun_out/r06/hiptrace_td_steady.txt
find $OUT -name "*.csv" -size +8M -delete
