#!/bin/bash
# round 6, lease 15: fc_out / fc0 weight gradients on the Python-side stream: parity + same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests -m gpu -q -k "encoder_vs_reference or fullsize or trajectory or ddp or direct_param or linear or encdec" > gpurun_out/r06/pytest_gpu_11.txt 2>&1; tail -3 gpurun_out/r06/pytest_gpu_11.txt | cut -c1-200
BENCH_ARGS="" bash tools/probes/r06_ab.sh linw "EA_LINEAR_WGRAD_SIDE=0" "EA_LINEAR_WGRAD_SIDE=1" | tee gpurun_out/r06/linear_wgrad_side_ab.txt
