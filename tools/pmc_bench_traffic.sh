#!/bin/bash
# HBM traffic of the dominant kernel (bf16 MFMA GEMMs) over bench.py's training step, per the MI355X guide: FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (TCC slots), kernel-trace only; gfx950 correction: FETCH_SIZE counts 128-B requests
# as 64 B -> doubled.  Writes gpurun_out/gemm_traffic.json (copy into profiles/ to have bench.py report it).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_traffic
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/$C -o $C --output-format csv -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-other-configs --no-roofline > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
python - <<PY
import csv, glob, json
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
n = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
for c in tot:
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and any(k in r["Kernel_Name"] for k in ("gemm_bf16_kernel", "gemm_glds_kernel", "gemm_w8_kernel", "wgrad_group_kernel", "wgrad_group_tr_kernel", "wgrad_w8_kernel")):
                tot[c] += float(r["Counter_Value"])
                n[c] += 1
assert n["FETCH_SIZE"] == n["WRITE_SIZE"] and n["FETCH_SIZE"] > 0, n
launches = n["FETCH_SIZE"]
read_b = 2.0 * tot["FETCH_SIZE"] * 1024.0 / launches   # gfx950: FETCH_SIZE reports half of a wide coalesced read
write_b = tot["WRITE_SIZE"] * 1024.0 / launches
out = {"kernel": "gemm_bf16_kernel + gemm_glds_kernel + gemm_w8_kernel + wgrad_w8_kernel + wgrad_group_tr_kernel (+ wgrad_group_kernel)", "launches": launches, "hbm_read_bytes_per_launch": read_b,
       "hbm_write_bytes_per_launch": write_b, "hbm_bytes_per_launch": read_b + write_b,
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 incl. the start-up reserve pass; "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)"}
json.dump(out, open("$R/gpurun_out/gemm_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
