#!/bin/bash
# round 6, lease 14: transposed weight copies refreshed off the compute stream: parity + same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests -m gpu -q -k "native_layer or fullsize or encoder_vs_reference or trajectory or ddp or deferred or checkpoint" > gpurun_out/r06/pytest_gpu_10.txt 2>&1; tail -3 gpurun_out/r06/pytest_gpu_10.txt | cut -c1-200
BENCH_ARGS="" bash tools/probes/r06_ab.sh wtpre "EA_WT_PREFETCH=0" "EA_WT_PREFETCH=1" | tee gpurun_out/r06/wt_prefetch_ab.txt
