cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or dropout_vs_restated or encoder_vs_reference or native_layer or layernorm" 2>&1 | grep -E "FAILED|passed|failed|^E |rror" | head -8
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
run() { (cd $1 && shift && env "$@" timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"); }
for i in 1 2 3; do
echo "== new (thin for >= 640 tiles)"; run . A=1
echo "== new EA_GEMM_THIN_MIN=0"; run . EA_GEMM_THIN_MIN=0
echo "== old"; run _old A=1
done
