#!/bin/bash
# round 6, lease 17: config 4 (transducer) with and without chained layer calls, same box, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
for rep in 1 2; do
  for E in "EA_LAYER_CHAIN=0" "EA_LAYER_CHAIN=1"; do
    env $E timeout 600 python tools/bench_transducer.py --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$E]', {k: round(d[k],3) for k in ('value','ms_per_step','host_enqueue_ms_per_step') if k in d})"
  done
done | tee gpurun_out/r06/transducer_chain_ab.txt
nproc; cat /proc/loadavg
