cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "ddp_every or direct_parameter" 2>&1 | grep -E "FAILED|passed|failed|^E " | head -10
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
run() { (cd $1 && shift && env "$@" python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['host_enqueue_ms_per_step'])"); }
for i in 1 2; do
echo "== new plain"; run . A=1
echo "== new DDP one rank, helper thread"; run . EA_DDP_FORCE=1
echo "== new DDP one rank, inline launches"; run . EA_DDP_FORCE=1 EA_DDP_THREAD=0
echo "== old DDP one rank"; run _old EA_DDP_FORCE=1
done
