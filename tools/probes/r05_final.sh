#!/bin/bash
# round 5, final tree: GPU suite + smoke, the full default bench line, the evidence run, a kernel trace of the one-rank RCCL mode
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
TAG=final bash tools/probes/r05_suite.sh
timeout 1500 python bench.py > gpurun_out/r05/bench_line_final.json 2> gpurun_out/r05/bench_line_final.err
tail -c 2500 gpurun_out/r05/bench_line_final.json
bash tools/probes/r05_evidence.sh 2>&1 | tail -60
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
bash $R/tools/probes/r05_ddp_trace.sh
