#!/bin/bash
# round 6, lease 3: fused joint + RNN-T loss (csrc/joint_rnnt.hip) — kernel parity, the transducer tests at the tightened bounds,
# config 4 fused vs materialised logits (same box, interleaved), kernel trace of the fused step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests -m gpu -q -k "joint or transducer or rnnt" > gpurun_out/r06/pytest_gpu_joint.txt 2>&1; tail -15 gpurun_out/r06/pytest_gpu_joint.txt | cut -c1-300
for rep in 1 2; do for F in 1 0; do
EA_JOINT_FUSED=$F timeout 600 python tools/bench_transducer.py --steps 8 > gpurun_out/r06/td_fused${F}_${rep}.json 2> gpurun_out/r06/td_fused${F}_${rep}.err
echo "EA_JOINT_FUSED=$F $(cut -c1-330 gpurun_out/r06/td_fused${F}_${rep}.json)"
done; done
