#!/bin/bash
# round 5, last lease: the two new trajectory tests + the full default bench line (reads the re-measured profiles/r05_gemm_traffic.json)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "encdec_training_trajectory or transducer_training_trajectory" > gpurun_out/r05/pytest_gpu_trajectories.txt 2>&1
grep -a "passed\|failed" gpurun_out/r05/pytest_gpu_trajectories.txt | tail -2
grep -a "max_rel_all" gpurun_out/r05/pytest_gpu_trajectories.txt | cut -c1-900 | tail -2
timeout 1500 python bench.py > gpurun_out/r05/bench_line_final2.json 2> gpurun_out/r05/bench_line_final2.err
tail -c 600 gpurun_out/r05/bench_line_final2.json
