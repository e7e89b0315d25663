cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-decode --no-other-configs --steps 20 --warmup 5"
for v in "" "--deferred-inline" "--no-bwd-overlap" ""; do
  echo "== $v"; python bench.py $F --no-roofline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
python bench.py $F --gemm-dump gpurun_out/r04_gemm_dump0.txt 2>/dev/null | tail -1 > gpurun_out/r04_bench0.json
python tools/gemm_shape_report.py gpurun_out/r04_gemm_dump0.txt > gpurun_out/r04_gemm_shapes0.txt 2>&1; head -50 gpurun_out/r04_gemm_shapes0.txt
