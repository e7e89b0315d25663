cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "deferred or encoder_vs_reference or fullsize_layer or native_layer or trajectory" 2>&1 | grep -E "FAILED|passed|failed|Error|assert" | head
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
run() { (cd $1 && shift && env "$@" python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"); }
for i in 1 2; do
echo "== new"; run . A=1
echo "== new EA_BN_UNFUSED"; run . EA_BN_UNFUSED=1
echo "== prev"; run _prev A=1
echo "== old"; run _old A=1
done
