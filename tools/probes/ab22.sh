cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "wgrad or deferred or native_layer or encoder_vs_reference" 2>&1 | grep -E "FAILED|passed|failed|^E |rror" | head -8
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5"
run() { (env "$@" python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['ms_per_step'],3), round(d['value'],3), 'frac',round(r['frac'],4),'gemm_ms',round(r['gemm_ms_per_step'],2),'avg_us',round(r['avg_launch_us'],1))"); }
echo "== default (thr 320)"; run A=1
echo "== EA_WGRAD_BM_THR=1024 (old rule)"; run EA_WGRAD_BM_THR=1024
echo "== encdec default / old rule"
python tools/bench_encdec.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d.get('value'))"
EA_WGRAD_BM_THR=1024 python tools/bench_encdec.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d.get('value'))"
echo "== transducer default / old rule"
python tools/bench_transducer.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d.get('value'))"
EA_WGRAD_BM_THR=1024 python tools/bench_transducer.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d.get('value'))"
