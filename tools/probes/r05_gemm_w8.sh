#!/bin/bash
# round 5: the 8-wave kernels against the 4-wave kernels and the vendor ceiling, per shape (parity flag in the table) + workgroup timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
TAG=${TAG:-v2}
timeout 200 tools/probes/gemm_w8_timing 6240 > gpurun_out/r05/w8_timing_$TAG.txt 2>&1
QUICK=1 timeout 900 python tools/bench_gemm_shapes.py 6240 > gpurun_out/r05/gemm_shapes_w8_$TAG.txt 2>&1
cut -c1-330 gpurun_out/r05/w8_timing_$TAG.txt | tail -12
cut -c1-250 gpurun_out/r05/gemm_shapes_w8_$TAG.txt | tail -16
