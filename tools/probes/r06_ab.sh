#!/bin/bash
# round 5: same-box A/B of the default bench step under environment switches.  usage: r06_ab.sh TAG "ENV1" "ENV2" ... (each ENV is a
# string of VAR=value pairs, "" = defaults); every variant runs twice, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
TAG=$1; shift
for rep in 1 2; do
  i=0
  for E in "$@"; do
    i=$((i+1))
    env $E timeout 600 python bench.py --no-cpu-baseline --no-decode --no-other-configs ${BENCH_ARGS} > gpurun_out/r06/ab_${TAG}_${i}_${rep}.json 2> gpurun_out/r06/ab_${TAG}_${i}_${rep}.err
    python - "$E" gpurun_out/r06/ab_${TAG}_${i}_${rep}.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(f"[{sys.argv[1]:40s}] ms_per_step {d['ms_per_step']:.3f}  gemm_ms {r.get('gemm_ms_per_step', 0):.2f}  frac {r.get('frac', 0):.4f}  loss {d['config']['last_loss_per_sentence']:.3f}")
except Exception as e:
    print(f"[{sys.argv[1]}] FAILED {e}")
PY
  done
done
