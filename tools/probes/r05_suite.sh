#!/bin/bash
# round 5: GPU suite + default bench line (+ the same bench with the 8-wave kernels off, same box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
TAG=${TAG:-a}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05/pytest_gpu_$TAG.txt 2>&1
tail -5 gpurun_out/r05/pytest_gpu_$TAG.txt
if [ "${BENCH:-1}" = "1" ]; then
  timeout 900 python bench.py --no-cpu-baseline --no-decode --no-other-configs > gpurun_out/r05/bench_$TAG.json 2> gpurun_out/r05/bench_$TAG.err
  tail -c 1500 gpurun_out/r05/bench_$TAG.json
  EA_GEMM_W8=0 timeout 900 python bench.py --no-cpu-baseline --no-decode --no-other-configs > gpurun_out/r05/bench_${TAG}_w8off.json 2> gpurun_out/r05/bench_${TAG}_w8off.err
  tail -c 1500 gpurun_out/r05/bench_${TAG}_w8off.json
fi
