#!/bin/bash
# round 6, lease 12: GLU / depthwise backward with its operand loads hoisted, conv-1 BatchNorm-backward + weight gradient with branch-free
# loads: parity tests, kernel trace, two bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -m gpu -q -k "conv or batchnorm or bn or native_layer or fullsize_layer or encoder_vs_reference or subsample" > gpurun_out/r06/pytest_gpu_8.txt 2>&1; tail -3 gpurun_out/r06/pytest_gpu_8.txt | cut -c1-200
bash tools/profile_bench.sh r06/prof_bench3 8 > gpurun_out/r06/prof_bench3.log 2>&1; grep -E "glu_dwconv_bwd_data|conv1_bn_bwd_wgrad|bn_act_bwd_reduce|per step" gpurun_out/r06/prof_bench3_summary.txt gpurun_out/r06/prof_bench3_gaps.txt | cut -c1-190
for rep in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-decode --no-other-configs > gpurun_out/r06/bench_glu_${rep}.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r06/bench_glu_${rep}.json')); print('bench', d['ms_per_step'], d['value'])"; done
