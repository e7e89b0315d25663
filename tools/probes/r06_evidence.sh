#!/bin/bash
# round 5 evidence (one lease): (a) kernel trace + gaps + streams of the default bench step, (b) MFMA-busy PMC pass over the same
# step, (c) kernel trace of the config-5 decode block (beam 10 + look-ahead word LM), (d) HBM traffic of the GEMM family
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
# (a)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_bench -o bench -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-decode --no-other-configs --no-roofline > $O/trace_bench.log 2>&1
DB=$(ls $O/trace_bench/*.db $O/trace_bench/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $O/r06_bench_kernel_trace.txt > /dev/null
python $R/tools/gap_analysis.py $DB 8 > $O/r06_gap_analysis.txt 2>&1
python $R/tools/stream_analysis.py $DB 8 > $O/r06_stream_analysis.txt 2>&1
head -30 $O/r06_bench_kernel_trace.txt | cut -c1-160
# (b)
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace -d $O/pmc_mfma_$i -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-other-configs --no-roofline > $O/pmc_mfma_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/pmc_mfma_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = None
        for k in ("gemm_w8_kernel", "gemm_glds_kernel", "gemm_bf16_kernel", "wgrad_group_tr_kernel", "rp_fwd_kernel", "rp_bwd_q_kernel", "rp_bwd_kv_kernel", "conv_gather_kernel", "conv_wgrad_kernel", "ln_bwd_kernel", "ln_fwd_rows_kernel"):
            if k in n:
                key = k
        if key:
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/r06_mfma_pmc.txt", "w") as o:
    print("# rocprofv3 --pmc over bench.py --steps 2 --warmup 1 (kernels serialised by the profiler): per-kernel means.", file=o)
    print("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES/32 shader engines x 1024 SIMDs) (busy cycles of the matrix pipes over the cycles the kernel kept the shader busy);", file=o)
    print("# alt = the same over GRBM_GUI_ACTIVE / 8 XCDs (includes dispatch overhead under the profiler)", file=o)
    for k, d in rows.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        busy = m.get("SQ_BUSY_CYCLES", 0) / 32.0
        gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
        u1 = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (busy * 1024) if busy else 0
        u2 = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024) if gui else 0
        print(f"{k:24s} launches {len(d.get('SQ_WAVES', []))}  mfma_util {u1:.3f}  alt {u2:.3f}  " + "  ".join(f"{c}={v:.0f}" for c, v in sorted(m.items())), file=o)
print(open("$O/r06_mfma_pmc.txt").read())
PY
# (c)
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_decode -o dec -- python $R/tools/bench_decode.py --wordlm --batches 6 > $O/trace_decode.log 2>&1
DB=$(ls $O/trace_decode/*.db $O/trace_decode/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $O/r06_decode_kernel_trace.txt > /dev/null
python $R/tools/gap_analysis.py $DB 1 > $O/r06_decode_gap_analysis.txt 2>&1
tail -1 $O/trace_decode.log | cut -c1-600
head -25 $O/r06_decode_kernel_trace.txt | cut -c1-160
head -8 $O/r06_decode_gap_analysis.txt
# (d)
bash $R/tools/pmc_bench_traffic.sh > $O/traffic.log 2>&1
tail -1 $O/traffic.log | cut -c1-600
cp $R/gpurun_out/gemm_traffic.json $O/r06_gemm_traffic.json 2>/dev/null
rm -rf $O/trace_bench $O/trace_decode $O/pmc_mfma_1 $O/pmc_mfma_2 $R/gpurun_out/pmc_traffic
