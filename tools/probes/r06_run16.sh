#!/bin/bash
# round 6, lease 16: chained Conformer layer calls (one LayerNorm kernel per layer boundary, both passes): parity + same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -q -x -k "chained or native_layer or deferred or encoder_dropout or encoder_vs_reference or fullsize or trajectory or ddp or layerdrop or transducer" > gpurun_out/r06/pytest_gpu_12.txt 2>&1; tail -5 gpurun_out/r06/pytest_gpu_12.txt | cut -c1-300
BENCH_ARGS="" bash tools/probes/r06_ab.sh chain "EA_LAYER_CHAIN=0" "EA_LAYER_CHAIN=1" | tee gpurun_out/r06/layer_chain_ab.txt
