# round-4 evidence run: PMC traffic, the full driver line, same-box A/B against the round-3 tree, kernel trace + analyses,
# isolated GEMM table, workgroup timing probe, the whole -m gpu suite.  Everything lands under gpurun_out/.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
bash tools/pmc_bench_traffic.sh > gpurun_out/r04_pmc.log 2>&1
cp gpurun_out/gemm_traffic.json profiles/r04_gemm_traffic.json 2>/dev/null
tail -2 gpurun_out/r04_pmc.log | cut -c1-400
cd $R
( time python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.err ) 2>&1 | grep real
tail -c 1500 gpurun_out/r04_bench_line.json | cut -c1-1500
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
run() { (cd $1 && shift && env "$@" python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"); }
for i in 1 2; do
echo "== new"; run . A=1
echo "== old (round-3 tree)"; run _old A=1
done
echo "== new, EA_WGRAD_BM_THR=1024 (the 64-row rule of round 3)"; run . EA_WGRAD_BM_THR=1024
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r04_trace_final
rm -rf $O; mkdir -p $O
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $O -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-other-configs --no-roofline > $O.log 2>&1)
DB=$(ls $O/*.db $O/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/r04_bench_kernel_trace.txt > /dev/null
python $R/tools/gemm_instep_report.py $DB 8 > $R/gpurun_out/r04_gemm_instep.txt 2>&1
python $R/tools/stream_analysis.py $DB 8 > $R/gpurun_out/r04_stream_analysis.txt 2>&1
python $R/tools/gap_analysis.py $DB 8 > $R/gpurun_out/r04_gap_analysis.txt 2>&1
head -12 $R/gpurun_out/r04_bench_kernel_trace.txt | cut -c1-160
cd $R
