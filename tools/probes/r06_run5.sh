#!/bin/bash
# round 6, lease 5: conv weight-gradient ring depth A/B, fused joint after the epilogue fixes (trace + step A/B), FLAC ingestion
# block, whole -m gpu suite, rocprofv3 trace / stream / gap analysis of the bench step, GEMM-family HBM traffic (PMC)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
for D in 1 3 1 3; do echo "EA_CONV_WGRAD_DEPTH=$D"; EA_CONV_WGRAD_DEPTH=$D python tools/probes/r06_conv_wgrad_time.py 2>/dev/null; done | tee gpurun_out/r06/conv_wgrad_depth_ab.txt
EA_JOINT_FUSED=1 bash tools/profile_transducer.sh r06/td_fused2 8 > gpurun_out/r06/td_fused2_prof.log 2>&1; grep -E "joint_rnnt|joint_fold|total kernel" gpurun_out/r06/td_fused2_summary.txt | cut -c1-170
for rep in 1 2; do for F in 1 0; do
EA_JOINT_FUSED=$F timeout 600 python tools/bench_transducer.py --steps 8 > gpurun_out/r06/td2_fused${F}_${rep}.json 2> /dev/null
echo "EA_JOINT_FUSED=$F $(python -c "import json,sys; d=json.load(open('gpurun_out/r06/td2_fused${F}_${rep}.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['peak_mem_gb'])")"
done; done | tee gpurun_out/r06/td_fused_ab.txt
timeout 1200 python tools/bench_ingest.py --format flac --files 768 > gpurun_out/r06/ingest_flac.json 2> gpurun_out/r06/ingest_flac.err; cut -c1-900 gpurun_out/r06/ingest_flac.json; tail -2 gpurun_out/r06/ingest_flac.err
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06/pytest_gpu_3.txt 2>&1; tail -6 gpurun_out/r06/pytest_gpu_3.txt | cut -c1-300
bash tools/profile_bench.sh r06/prof_bench 12 > gpurun_out/r06/prof_bench.log 2>&1; tail -30 gpurun_out/r06/prof_bench.log | cut -c1-200
bash tools/pmc_bench_traffic.sh > gpurun_out/r06/pmc_traffic.log 2>&1; tail -2 gpurun_out/r06/pmc_traffic.log | cut -c1-600; cp gpurun_out/gemm_traffic.json gpurun_out/r06/ 2>/dev/null
