cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5"
run() { (env "$@" python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['ms_per_step'],3), round(d['value'],3), 'frac',round(r['frac'],4),'gemm_ms',round(r['gemm_ms_per_step'],2),'avg_us',round(r['avg_launch_us'],1))"); }
for i in 1 2; do
echo "== BM auto"; run A=1
echo "== EA_WGRAD_BM=128"; run EA_WGRAD_BM=128
done
echo "== encdec auto / 128"
python tools/bench_encdec.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d.get('value'))"
EA_WGRAD_BM=128 python tools/bench_encdec.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d.get('value'))"
