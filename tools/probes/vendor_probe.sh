# one-off (round 4): which kernels the vendor library picks for the hot GEMM shapes (name, grid, LDS, VGPRs), and how long the
# grouped weight-gradient kernel takes when nothing runs next to it (bench.py --deferred-inline)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_vendor
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O -o blas -- python $R/tools/bench_blas_reference.py > $O.log 2>&1
DB=$(ls $O/*.db $O/*/*.db 2>/dev/null | head -1)
python - <<PY
import sqlite3
db = sqlite3.connect("$DB"); cur = db.cursor()
kv = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith('kernels')][0]
rows = cur.execute(f"select name, count(*), avg(end-start)/1e3, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, accum_vgpr_count from {kv} group by name, grid_x, grid_y order by 3 desc").fetchall()
for r in rows:
    if r[1] >= 20: print(f"{r[2]:8.1f} us x{r[1]:4d} grid {r[3]}x{r[4]}x{r[5]} wg {r[6]} lds {r[7]} vgpr {r[8]}+{r[9]}  {r[0][:150]}")
PY
O2=$R/gpurun_out/r04_inline
mkdir -p $O2
timeout 600 rocprofv3 --kernel-trace -d $O2 -o bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-other-configs --no-roofline --deferred-inline > $O2.log 2>&1
DB=$(ls $O2/*.db $O2/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/r04_inline_summary.txt > /dev/null
grep -E "wgrad_group|gemm_bf16_kernel<false, false, 64>|total kernel" $R/gpurun_out/r04_inline_summary.txt | cut -c1-160
tail -1 $O2.log | cut -c1-200
