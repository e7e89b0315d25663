#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
EA_CONV_WGRAD_TAPS=3 timeout 600 python -m pytest tests -m gpu -q -k "conv" 2>&1 | tail -2 | cut -c1-200
for T in 9 3 9 3; do echo "EA_CONV_WGRAD_TAPS=$T: $(EA_CONV_WGRAD_TAPS=$T python tools/probes/r06_conv_wgrad_time.py 2>/dev/null | awk '{printf "%s %s us  ", $1, $2}')"; done | tee gpurun_out/r06/convw_taps_ab.txt
BENCH_ARGS="" bash tools/probes/r06_ab.sh convtaps "EA_CONV_WGRAD_TAPS=9" "EA_CONV_WGRAD_TAPS=3" | tee -a gpurun_out/r06/convw_taps_ab.txt
