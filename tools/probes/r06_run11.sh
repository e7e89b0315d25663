#!/bin/bash
# round 6, lease 11: BatchNorm backward reduce with four rows in flight: parity + kernel time; then the round's evidence set
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -m gpu -q -k "conv or batchnorm or bn or native_layer or fullsize_layer or encoder_vs_reference" > gpurun_out/r06/pytest_gpu_7.txt 2>&1; tail -3 gpurun_out/r06/pytest_gpu_7.txt | cut -c1-200
bash tools/probes/r06_evidence.sh > gpurun_out/r06/evidence.log 2>&1
grep -E "bn_act_bwd_reduce|ln_bwd_kernel|ctc_scan" gpurun_out/r06/r06_bench_kernel_trace.txt | cut -c1-150
head -14 gpurun_out/r06/r06_mfma_pmc.txt | cut -c1-120
for rep in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-decode --no-other-configs > gpurun_out/r06/bench_bn_${rep}.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r06/bench_bn_${rep}.json')); print('bench', d['ms_per_step'], d['value'])"; done
