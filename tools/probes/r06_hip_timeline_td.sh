#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/tl
mkdir -p $OUT $R/gpurun_out/r06
EA_LAYER_STACK=${EA_LAYER_STACK:-1} timeout 900 rocprofv3 --hip-runtime-trace --kernel-trace -d $OUT -o td -f csv -- python $R/tools/bench_transducer.py --steps 6 --warmup 3 > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-200
A=$(ls $OUT/*hip_api_trace.csv $OUT/*/*hip_api_trace.csv 2>/dev/null | head -1)
K=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/probes/hip_timeline.py $A $K 260 0.85 > $R/gpurun_out/r06/hip_timeline_td.txt 2>&1
head -140 $R/gpurun_out/r06/hip_timeline_td.txt
