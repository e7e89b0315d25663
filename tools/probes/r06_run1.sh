#!/bin/bash
# round 6, lease 1: fp32 joint — transducer parity numbers, then the whole -m gpu suite and a bench baseline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
python - > gpurun_out/r06/joint_f32_checks.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, ".")
from tests import gpu_checks as G
for rep in range(3):
    r = G.check_transducer_vs_reference()
    print("vs_reference", rep, json.dumps({k: r[k] for k in ("eval_logits_abs", "train_logits_abs", "worst_l2", "worst_scale", "train_logits_vs_emulation", "worst_l2_vs_emulation", "median_l2_vs_emulation", "fc_out_max") if k in r}, default=str))
r = G.check_transducer_dropout_vs_oracle()
print("dropout", json.dumps({k: r[k] for k in ("train_logits_vs_emulation", "worst_l2_vs_emulation", "median_l2_vs_emulation", "wrong_mask_median_l2") if k in r}, default=str))
r = G.check_transducer_training_trajectory()
print("trajectory", json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != "losses"}) for k, v in r.items() if k != "hip_losses"}, default=str))
PY
tail -8 gpurun_out/r06/joint_f32_checks.txt | cut -c1-600
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06/pytest_gpu_1.txt 2>&1; tail -5 gpurun_out/r06/pytest_gpu_1.txt
timeout 600 python bench.py --no-cpu-baseline --no-decode --no-other-configs > gpurun_out/r06/bench_base.json 2> gpurun_out/r06/bench_base.err; cut -c1-400 gpurun_out/r06/bench_base.json
