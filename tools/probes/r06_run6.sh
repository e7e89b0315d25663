#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 600 python tools/probes/r06_aten_ops.py > gpurun_out/r06/aten_ops.txt 2>gpurun_out/r06/aten_ops.err; head -75 gpurun_out/r06/aten_ops.txt; tail -3 gpurun_out/r06/aten_ops.err
EA_JOINT_FUSED=1 bash tools/profile_transducer.sh r06/td_fused3 8 > gpurun_out/r06/td_fused3_prof.log 2>&1; grep -E "joint_rnnt|total kernel" gpurun_out/r06/td_fused3_summary.txt | cut -c1-170
for rep in 1 2; do for F in 1 0; do
EA_JOINT_FUSED=$F timeout 600 python tools/bench_transducer.py --steps 8 > gpurun_out/r06/td3_fused${F}_${rep}.json 2> /dev/null
echo "EA_JOINT_FUSED=$F $(python -c "import json,sys; d=json.load(open('gpurun_out/r06/td3_fused${F}_${rep}.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['peak_mem_gb'])")"
done; done | tee gpurun_out/r06/td_fused_ab3.txt
timeout 900 python -m pytest tests -m gpu -q -k "joint or transducer or conv" > gpurun_out/r06/pytest_gpu_4.txt 2>&1; tail -4 gpurun_out/r06/pytest_gpu_4.txt | cut -c1-300
