#!/bin/bash
# round 5: kernel trace + stream analysis of the bench step with the data-parallel wrapper on a one-rank RCCL group
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
EA_DDP_FORCE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_ddp -o ddp -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-other-configs --no-roofline > $O/trace_ddp.log 2>&1
DB=$(ls $O/trace_ddp/*.db $O/trace_ddp/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $O/r05_ddp_one_rank_kernel_trace.txt > /dev/null
python $R/tools/stream_analysis.py $DB 6 > $O/r05_ddp_one_rank_stream_analysis.txt 2>&1
grep -i "nccl\|rccl" $O/r05_ddp_one_rank_kernel_trace.txt | head -5 | cut -c1-200
sed -n 2,8p $O/r05_ddp_one_rank_stream_analysis.txt
rm -rf $O/trace_ddp
