cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "conv3x3 or conv_subsample or subsample or encoder_vs_reference or dropout" 2>&1 | grep -E "FAILED|passed|failed|^E " | head -8
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
run() { (cd $1 && shift && env "$@" python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"); }
for i in 1 2 3; do
echo "== new"; run . A=1
echo "== old"; run _old A=1
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_trace
rm -rf $O; mkdir -p $O
(cd $R && timeout 600 rocprofv3 --kernel-trace -d $O -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-other-configs --no-roofline > $O.log 2>&1)
DB=$(ls $O/*.db $O/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/r04_kernel_summary.txt > /dev/null
head -45 $R/gpurun_out/r04_kernel_summary.txt
python $R/tools/gemm_instep_report.py $DB 8 > $R/gpurun_out/r04_gemm_instep.txt 2>&1; cat $R/gpurun_out/r04_gemm_instep.txt
python $R/tools/stream_analysis.py $DB 8 > $R/gpurun_out/r04_stream_analysis.txt 2>&1; tail -30 $R/gpurun_out/r04_stream_analysis.txt
