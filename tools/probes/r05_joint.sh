#!/bin/bash
# round 5: transducer joint GEMMs — parity tests of the new kernels, the isolated probe, config 4 with and without them (same box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wgrad or joint or 8wave or transducer" 2>&1 | tail -8 | tee gpurun_out/r05/pytest_joint.txt
timeout 600 python tools/probes/joint_gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/joint_gemm_probe.txt
for rep in 1 2; do
  for E in "EA_WGRAD_W8=1 EA_GEMM_W8_MANY=1024" "EA_WGRAD_W8=0 EA_GEMM_W8_MANY=100000000" "EA_WGRAD_W8=1 EA_GEMM_W8_MANY=100000000" "EA_WGRAD_W8=0 EA_GEMM_W8_MANY=1024"; do
    echo "[$E] rep $rep"
    env $E timeout 600 python tools/bench_transducer.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k in ('value','ms_per_step','loss','last_loss')})"
  done
done 2>&1 | tee gpurun_out/r05/joint_ab.txt
