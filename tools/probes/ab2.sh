cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5 --no-roofline"
rocm-smi --showclocks --showpower 2>/dev/null | head -30
for d in . _old . _old; do
  echo "== $d"; (cd $d && python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
done
