cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "flash_attention or dropout_vs_oracle" 2>&1 | grep -E "FAILED|passed|failed|Error|^E " | head -10
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
run() { (cd $1 && shift && env "$@" python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"); }
for i in 1 2; do
echo "== new (512)"; run . A=1
echo "== new uncapped"; run . EA_KEEP_BITS_WGS=0
echo "== new 256"; run . EA_KEEP_BITS_WGS=256
echo "== new 1024"; run . EA_KEEP_BITS_WGS=1024
echo "== old"; run _old A=1
done
