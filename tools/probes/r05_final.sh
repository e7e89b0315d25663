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
EA_DDP_FORCE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_ddp -o ddp -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-other-configs --no-roofline > $O/trace_ddp.log 2>&1
DB=$(ls $O/trace_ddp/*.db $O/trace_ddp/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $O/r05_ddp_one_rank_kernel_trace.txt > /dev/null
python $R/tools/stream_analysis.py $DB 6 > $O/r05_ddp_one_rank_stream_analysis.txt 2>&1
grep -i "nccl\|rccl\|copyBuffer" $O/r05_ddp_one_rank_kernel_trace.txt | head -5 | cut -c1-200
head -8 $O/r05_ddp_one_rank_stream_analysis.txt
rm -rf $O/trace_ddp
