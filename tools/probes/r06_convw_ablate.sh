#!/bin/bash
# round 6: what bounds conv_wgrad_kernel?  Ablation by runtime switches (results are meaningless, only the time counts), both ring depths
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
for D in 1 3; do for A in 0 1 2 4 8 3 5 6 7 15; do
echo "depth $D ablate $A: $(EA_CONV_WGRAD_DEPTH=$D EA_CONVW_ABLATE=$A python tools/probes/r06_conv_wgrad_time.py 2>/dev/null | awk '{printf "%s %s us  ", $1, $2}')"
done; done | tee gpurun_out/r06/convw_ablate.txt
