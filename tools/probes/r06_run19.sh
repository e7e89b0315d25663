#!/bin/bash
# round 6, lease 19: 8-wave weight-gradient kernel on the layer groups of config 4 / config 2 (forced), and parity of the wgrad tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
for rep in 1 2; do
  for E in "EA_WGRAD_W8=1" "EA_WGRAD_W8=2"; do
    env $E timeout 600 python tools/bench_transducer.py --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[config 4, $E]', {k: round(d[k],3) for k in ('value','ms_per_step','host_enqueue_ms_per_step') if k in d})"
    env $E timeout 600 python tools/bench_encdec.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[config 2, $E]', {k: round(d[k],3) for k in ('value','ms_per_step') if k in d})"
  done
done | tee gpurun_out/r06/wgrad_w8_configs_ab.txt
timeout 900 python -m pytest tests -m gpu -q -k "wgrad or conv or subsamp or direct_param" 2>&1 | tail -3
