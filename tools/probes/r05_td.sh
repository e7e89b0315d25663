#!/bin/bash
# round 5: transducer tests + config-4 step + kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rnnt or transducer or joint" 2>&1 | tail -4
for rep in 1 2; do
  timeout 600 python tools/bench_transducer.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k in ('value','ms_per_step','host_enqueue_ms_per_step')})"
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
timeout 600 rocprofv3 --kernel-trace --stats -d $O/td_v1 -o td -- python $R/tools/bench_transducer.py --steps 6 --warmup 2 > $O/td_v1.log 2>&1
DB=$(ls $O/td_v1/*.db $O/td_v1/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $O/td_v1_summary.txt > /dev/null
python $R/tools/gap_analysis.py $DB 4 > $O/td_v1_gaps.txt 2>&1
head -24 $O/td_v1_summary.txt | cut -c1-150
head -3 $O/td_v1_gaps.txt
rm -rf $O/td_v1
