#!/bin/bash
# round 6, lease 7: scan kernels with LDS-only barriers + three-deep prefetch (ctc_scan, rnnt_scan): parity tests, kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -m gpu -q -k "ctc or rnnt or transducer or joint or loss" > gpurun_out/r06/pytest_gpu_5.txt 2>&1; tail -4 gpurun_out/r06/pytest_gpu_5.txt | cut -c1-300
bash tools/profile_bench.sh r06/prof_bench2 8 > gpurun_out/r06/prof_bench2.log 2>&1; grep -E "ctc_scan|ctc_grad|log_softmax|per step" gpurun_out/r06/prof_bench2_summary.txt gpurun_out/r06/prof_bench2_gaps.txt | cut -c1-200
EA_JOINT_FUSED=1 bash tools/profile_transducer.sh r06/td_fused4 8 > gpurun_out/r06/td_fused4_prof.log 2>&1; grep -E "joint_rnnt|rnnt_scan|total kernel" gpurun_out/r06/td_fused4_summary.txt | cut -c1-170
for rep in 1 2; do
timeout 600 python tools/bench_transducer.py --steps 8 > gpurun_out/r06/td4_${rep}.json 2> /dev/null
echo "td $(python -c "import json,sys; d=json.load(open('gpurun_out/r06/td4_${rep}.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])")"
timeout 600 python bench.py --no-cpu-baseline --no-decode --no-other-configs > gpurun_out/r06/bench_scan_${rep}.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r06/bench_scan_${rep}.json')); print('bench', d['ms_per_step'], d['value'])"
done
