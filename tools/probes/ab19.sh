cd $GRAFT_REPO_ROOT
timeout 120 tools/probes/gemm_timing 6240 | cut -c1-235
echo "--- parity with the thin kernel forced (EA_GEMM_GLDS=11)"
EA_GEMM_GLDS=11 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or dropout_vs_restated or encoder_vs_reference or native_layer" 2>&1 | grep -E "FAILED|passed|failed|^E |rror" | head -8
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
run() { (cd $1 && shift && env "$@" timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"); }
for i in 1 2; do
echo "== new auto"; run . A=1
echo "== new EA_GEMM_GLDS=11"; run . EA_GEMM_GLDS=11
echo "== old"; run _old A=1
done
