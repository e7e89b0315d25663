#!/bin/bash
# round 6, lease 20: second clause of the 8-wave weight-gradient rule (>= 80 tiles and >= 1024 rows) on configs 4, 2 (and 3 unchanged)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
for rep in 1 2; do
  for E in "EA_WGRAD_W8_SMALL_ROWS_TILES=0" "EA_WGRAD_W8_SMALL_ROWS_TILES=80"; do
    env $E timeout 600 python tools/bench_transducer.py --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[config 4, $E]', {k: round(d[k],3) for k in ('value','ms_per_step','host_enqueue_ms_per_step') if k in d})"
    env $E timeout 600 python tools/bench_encdec.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[config 2, $E]', {k: round(d[k],3) for k in ('value','ms_per_step') if k in d})"
  done
done | tee gpurun_out/r06/wgrad_w8_small_rows_ab.txt
