cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "direct_parameter or ctc_fp32 or fullsize_ctc or fullsize_encdec or encoder_vs_reference or deferred or trajectory or native_layer or ddp_every" 2>&1 | grep -E "FAILED|passed|failed|Error|^E " | head -20
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
run() { (cd $1 && shift && env "$@" python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"); }
for i in 1 2; do
echo "== new"; run . A=1
echo "== new EA_DIRECT_GRADS=0"; run . EA_DIRECT_GRADS=0
echo "== prev"; run _prev A=1
echo "== old"; run _old A=1
done
