cd $GRAFT_REPO_ROOT
timeout 120 tools/probes/gemm_timing 6240 | cut -c1-250
echo "--- parity with the B-direct kernel forced (EA_GEMM_GLDS=13, 14)"
EA_GEMM_GLDS=13 python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or dropout_vs_restated or encoder_vs_reference or native_layer" 2>&1 | grep -E "FAILED|passed|failed|^E " | head -8
EA_GEMM_GLDS=14 python -m pytest tests/test_gpu_parity.py -x -q -k "gemm" 2>&1 | grep -E "FAILED|passed|failed|^E " | head -8
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
run() { (cd $1 && shift && env "$@" python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"); }
for i in 1 2; do
echo "== new auto"; run . A=1
echo "== new EA_GEMM_GLDS=13"; run . EA_GEMM_GLDS=13
echo "== new EA_GEMM_GLDS=14"; run . EA_GEMM_GLDS=14
echo "== old"; run _old A=1
done
