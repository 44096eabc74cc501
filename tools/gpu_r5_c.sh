#!/bin/bash
# round 5, GPU call C: K1b variants 3 / 4 / 5 (exact rule; 4 / 2 / 1 tasks per turn of the cooperative copy), SQ counters of variants 1 and 3,
# two half-size passes side by side (does the device overlap the kernels of two slices?), the driver's invocation with the new accounting.
set -u
OUT=gpurun_out/r5_c
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
for v in 4 5; do
  SBX_K1B_VARIANT=$v timeout 200 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_depth.py -x -q 2>&1 | tail -3 | tee $OUT/tests_k1b_variant$v.txt
done
show() {
python - $1 <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], "Mreads/s", d["value"], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, "parity", d["parity_checked"]["ok"], d["parity_checked"].get("text_ok"), (d["parity_checked"].get("full_text") or {}).get("coverage"))
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
for v in 3 4 5; do
  SBX_K1B_VARIANT=$v timeout 120 python bench.py --length 40000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 --no-full-parity > $OUT/bench_40Mbp_k1b$v.json 2> /tmp/b40_$v.err
  show $OUT/bench_40Mbp_k1b$v.json
done
for v in 1 3 4; do
  ( cd /tmp && SBX_K1B_VARIANT=$v timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d /tmp/sq_v$v -o s -- \
      python $REPO/bench.py --length 40000000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 --no-full-parity > /dev/null 2> /tmp/sq_v$v.err )
  python - $v <<'PY' | tee -a $OUT/sq_insts_40Mbp.txt
import csv, glob, collections, re, sys
acc = collections.defaultdict(float)
for f in glob.glob("/tmp/sq_v%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", row["Kernel_Name"])
        acc[(m.group(1) if m else row["Kernel_Name"][:30], row["Counter_Name"])] += float(row["Counter_Value"])
for (k, c), v in sorted(acc.items()):
    if v > 1e6 and ("lz77" in k or "describe" in k): print("variant", sys.argv[1], k, c, "%.4g" % v)
PY
done
# two half-size passes side by side
H=124478211
timeout 300 python bench.py --length $H --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --parity-windows 2 --no-full-parity > $OUT/bench_half_alone.json 2> /tmp/half0.err; show $OUT/bench_half_alone.json
( timeout 300 python bench.py --length $H --steps 60 --warmup 3 --no-cpu-baseline --no-e2e --parity-windows 0 --no-full-parity > $OUT/bench_half_side_a.json 2> /tmp/half1.err ) &
( timeout 300 python bench.py --length $H --steps 60 --warmup 3 --no-cpu-baseline --no-e2e --parity-windows 0 --no-full-parity > $OUT/bench_half_side_b.json 2> /tmp/half2.err ) &
wait
show $OUT/bench_half_side_a.json; show $OUT/bench_half_side_b.json
# the driver's invocation
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_config2_driver_invocation.json 2> $OUT/bench_config2_driver_invocation.err; show $OUT/bench_config2_driver_invocation.json; tail -12 $OUT/bench_config2_driver_invocation.err
