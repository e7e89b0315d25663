#!/bin/bash
# round 6, final validation: smoke, whole -m gpu suite, the default bench line (what the driver runs), kernel trace / stream / gap analysis
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06/r06_smoke.txt 2>&1; tail -2 gpurun_out/r06/r06_smoke.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06/r06_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r06/r06_pytest_gpu.txt | cut -c1-300
timeout 900 python bench.py > gpurun_out/r06/r06_bench_line.json 2> gpurun_out/r06/r06_bench_line.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/r06_bench_line.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["config"]["host_enqueue_ms_fastest_step"], d["config"].get("side_stream_queue_probe_per_rank"))
for k in ("ingest", "ingest_flac", "config2_encdec", "config4_transducer", "f3_transducer_beam_search"):
    b = d.get(k) or {}
    print(k, {kk: b.get(kk) for kk in ("value", "ms_per_step", "end_to_end_over_resident", "rtf", "error") if kk in b})
print("decode", {kk: d["decode"].get(kk) for kk in ("rtf", "wall_seconds")}, "cpu", d["cpu_baseline"]["value"])
PY
bash tools/profile_bench.sh r06/prof_final 12 > gpurun_out/r06/prof_final.log 2>&1; head -4 gpurun_out/r06/prof_final_streams.txt | tail -3
