#!/bin/bash
# round 6, lease 2: where the remaining transducer gradient noise comes from (fp32 lattice logits, diagnostic), the whole -m gpu suite,
# keep-bits inline vs side-stream kernel (same box, interleaved), config 4 host profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
for F32 in 0 1; do
EA_JOINT_LOGITS_F32=$F32 python - >> gpurun_out/r06/joint_logits_f32_ab.txt 2>/dev/null <<'PY'
import json, os, sys
sys.path.insert(0, ".")
from tests import gpu_checks as G
r = G.check_transducer_vs_reference()
print("EA_JOINT_LOGITS_F32=" + os.environ["EA_JOINT_LOGITS_F32"], json.dumps({k: r[k] for k in ("eval_logits_abs", "train_logits_abs", "worst_l2", "train_logits_vs_emulation", "worst_l2_vs_emulation", "median_l2_vs_emulation") if k in r}, default=str))
PY
done
cat gpurun_out/r06/joint_logits_f32_ab.txt | cut -c1-500
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06/pytest_gpu_2.txt 2>&1; tail -8 gpurun_out/r06/pytest_gpu_2.txt
BENCH_ARGS="" bash tools/probes/r06_ab.sh bits "" "EA_FLASH_BITS_INLINE=1"
EA_TD_HOST_PROFILE=1 timeout 600 python tools/bench_transducer.py --steps 8 > gpurun_out/r06/td_hostprof.json 2> gpurun_out/r06/td_hostprof.err; cut -c1-600 gpurun_out/r06/td_hostprof.json
cp gpurun_out/host_profile_td.txt gpurun_out/r06/ 2>/dev/null
timeout 600 python tools/bench_transducer.py --steps 8 > gpurun_out/r06/td_base.json 2> gpurun_out/r06/td_base.err; cut -c1-600 gpurun_out/r06/td_base.json
