cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "wgrad or joint or transducer_training_mode or deferred" 2>&1 | grep -E "FAILED|passed|failed|^E |rror" | head -8
for i in 1 2; do
python tools/bench_transducer.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('transducer', d.get('ms_per_step'), d.get('value'))"
done
EA_WGRAD_BM_THR=1024 python tools/bench_transducer.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('transducer, 64-row rule', d.get('ms_per_step'), d.get('value'))"
