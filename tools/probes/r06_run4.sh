#!/bin/bash
# round 6, lease 4: kernel traces of config 4 with the fused / materialised joint, sub-sampler weight-gradient ring, logit-error
# table, FLAC end-to-end ingestion block
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
python tools/probes/r06_conv_wgrad_time.py 2>&1 | tee gpurun_out/r06/conv_wgrad_time.txt
timeout 900 python -m pytest tests -m gpu -q -k "conv" > gpurun_out/r06/pytest_gpu_conv.txt 2>&1; tail -4 gpurun_out/r06/pytest_gpu_conv.txt | cut -c1-300
EA_JOINT_FUSED=1 bash tools/profile_transducer.sh r06/td_fused 8 > gpurun_out/r06/td_fused_prof.log 2>&1; grep -E "joint|rnnt|gemm_w8|glds_kernel<128|wgrad_w8|total kernel" gpurun_out/r06/td_fused_summary.txt | cut -c1-170
EA_JOINT_FUSED=0 bash tools/profile_transducer.sh r06/td_unfused 8 > gpurun_out/r06/td_unfused_prof.log 2>&1; grep -E "joint|rnnt|gemm_w8|glds_kernel<128|wgrad_w8|total kernel" gpurun_out/r06/td_unfused_summary.txt | cut -c1-170
timeout 900 python tools/probes/r06_logit_errors.py > gpurun_out/r06/logit_errors.txt 2>/dev/null; cat gpurun_out/r06/logit_errors.txt | cut -c1-400
timeout 1200 python tools/bench_ingest.py --format flac --files 768 > gpurun_out/r06/ingest_flac.json 2> gpurun_out/r06/ingest_flac.err; cut -c1-700 gpurun_out/r06/ingest_flac.json; tail -3 gpurun_out/r06/ingest_flac.err
timeout 600 python bench.py --no-cpu-baseline --no-decode --no-other-configs > gpurun_out/r06/bench_wgrad_ring.json 2>/dev/null; cut -c1-300 gpurun_out/r06/bench_wgrad_ring.json
