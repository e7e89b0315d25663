#!/bin/bash
# round 5: GPU suite (+ smoke numbers) on the current tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
TAG=${TAG:-a}
timeout 1800 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > gpurun_out/r05/pytest_gpu_$TAG.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r05/pytest_gpu_$TAG.txt | tail -3
grep -B5 -A40 "FAILED\|Error" gpurun_out/r05/pytest_gpu_$TAG.txt | head -120
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke_$TAG.txt 2>&1
tail -5 gpurun_out/r05/smoke_$TAG.txt | cut -c1-1500
