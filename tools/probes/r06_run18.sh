#!/bin/bash
# round 6, lease 18: the layer loop as one C call per direction (ea_conformer_stack_fwd / _bwd): parity + same-box A/B on configs 3 and 4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -q -x -k "stack or chained or native_layer or deferred or encoder_dropout or encoder_vs_reference or fullsize or trajectory or ddp or layerdrop or transducer" > gpurun_out/r06/pytest_gpu_15.txt 2>&1; tail -5 gpurun_out/r06/pytest_gpu_15.txt | cut -c1-300
for rep in 1 2; do
  for E in "EA_LAYER_STACK=0" "EA_LAYER_STACK=1"; do
    env $E timeout 600 python tools/bench_transducer.py --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[config 4, $E]', {k: round(d[k],3) for k in ('value','ms_per_step','host_enqueue_ms_per_step') if k in d})"
  done
done | tee gpurun_out/r06/layer_stack_ab.txt
BENCH_ARGS="" bash tools/probes/r06_ab.sh stack "EA_LAYER_STACK=0" "EA_LAYER_STACK=1" | tee -a gpurun_out/r06/layer_stack_ab.txt
cat /proc/loadavg
