#!/bin/bash
# round 5: per-launch GEMM records of the bench step's roofline replay, default vs 8-wave kernels off (same box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
python bench.py --no-cpu-baseline --no-decode --no-other-configs --gemm-dump gpurun_out/r05/gemm_dump_w8.txt > /dev/null 2>&1
EA_GEMM_W8=0 python bench.py --no-cpu-baseline --no-decode --no-other-configs --gemm-dump gpurun_out/r05/gemm_dump_w8off.txt > /dev/null 2>&1
python tools/gemm_shape_report.py gpurun_out/r05/gemm_dump_w8.txt > gpurun_out/r05/gemm_shapes_instep_w8.txt
python tools/gemm_shape_report.py gpurun_out/r05/gemm_dump_w8off.txt > gpurun_out/r05/gemm_shapes_instep_w8off.txt
head -40 gpurun_out/r05/gemm_shapes_instep_w8.txt
