#!/bin/bash
# round 6, lease 13: GLU / depthwise kernels with 16-byte I/O through LDS: parity, kernel times, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests -m gpu -q -k "conv or native_layer or fullsize_layer or encoder_vs_reference or encoder_training_mode or trajectory or transducer_vs_reference" > gpurun_out/r06/pytest_gpu_9.txt 2>&1; tail -3 gpurun_out/r06/pytest_gpu_9.txt | cut -c1-200
bash tools/profile_bench.sh r06/prof_bench4 8 > gpurun_out/r06/prof_bench4.log 2>&1; grep -E "glu_dwconv|conv_wgrad_kernel|per step" gpurun_out/r06/prof_bench4_summary.txt gpurun_out/r06/prof_bench4_gaps.txt | cut -c1-190
for rep in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-decode --no-other-configs > gpurun_out/r06/bench_glu2_${rep}.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r06/bench_glu2_${rep}.json')); print('bench', d['ms_per_step'], d['value'])"; done
