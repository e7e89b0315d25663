#!/bin/bash
# round 6, lease 10: LayerNorm backward with two rows in flight (A/B), parity of the LayerNorm / layer tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -m gpu -q -k "layernorm or layer_norm or native_layer or fullsize_layer or ln_" > gpurun_out/r06/pytest_gpu_6.txt 2>&1; tail -3 gpurun_out/r06/pytest_gpu_6.txt | cut -c1-200
BENCH_ARGS="" bash tools/probes/r06_ab.sh lnbwd "EA_LN_BWD_TWO_ROWS=0" "EA_LN_BWD_TWO_ROWS=1" | tee gpurun_out/r06/ln_bwd_two_rows_ab.txt
