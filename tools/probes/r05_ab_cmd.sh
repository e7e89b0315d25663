#!/bin/bash
# round 5: same-box A/B of an arbitrary bench script under environment switches.  usage: r05_ab_cmd.sh TAG "python tools/x.py" "ENV1" "ENV2" ...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
TAG=$1; CMD=$2; shift; shift
for rep in 1 2; do
  for E in "$@"; do
    env $E timeout 600 $CMD 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[%-55s] ms_per_step %.3f value %.4f' % (sys.argv[1], d['ms_per_step'], d['value']))" "$E"
  done
done | tee gpurun_out/r05/abcmd_$TAG.txt
