cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "adam or optim or trajectory or flat" 2>&1 | grep -E "FAILED|passed|failed|^E " | head -8
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
run() { (cd $1 && shift && env "$@" python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"); }
for i in 1 2 3; do
echo "== new"; run . A=1
echo "== old"; run _old A=1
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_trace2
rm -rf $O; mkdir -p $O
(cd $R && timeout 600 rocprofv3 --kernel-trace -d $O -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-other-configs --no-roofline > $O.log 2>&1)
DB=$(ls $O/*.db $O/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/r04_kernel_summary2.txt > /dev/null
grep -E "adam|sumsq|gemm_glds|gemm_bf16_kernel<false, false, 64, true|ln_bwd|total kernel" $R/gpurun_out/r04_kernel_summary2.txt | cut -c1-150
