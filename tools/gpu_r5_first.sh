#!/bin/bash
# The first GPU call of the next round: K1b variant 2 (origin-pointer resolve, inflate.hip kJump -- written at the end of round 4,
# never run on a device).  1. does it inflate correctly (the inflate tests and the depth fixtures with the variant forced);
# 2. what is it worth (40 Mbp, then the full config 2), each against variant 1 in the same call.
# Raw output under /tmp on the box; only the summaries go to gpurun_out/ (64 MiB limit).
set -u
OUT=gpurun_out/r5_first
mkdir -p $OUT
SBX_K1B_VARIANT=2 timeout 240 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_depth.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -8 | tee $OUT/tests_variant2.txt
for v in 1 2; do
  SBX_K1B_VARIANT=$v timeout 120 python bench.py --length 40000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 12 > $OUT/bench_40Mbp_k1b_variant$v.json 2> /tmp/bench40_$v.err
  python - $OUT/bench_40Mbp_k1b_variant$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], "Mreads/s", d["value"], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, "parity", d["parity_checked"]["ok"], d["parity_checked"].get("text_ok"))
PY
done
for h in 4096 8192; do
  SBX_K1B_VARIANT=2 SBX_K1B_HIST=$h timeout 120 python bench.py --length 40000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 12 > $OUT/bench_40Mbp_k1b_variant2_hist$h.json 2> /tmp/bench40_h$h.err
  python - $OUT/bench_40Mbp_k1b_variant2_hist$h.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], "Mreads/s", d["value"], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, "parity", d["parity_checked"]["ok"], d["parity_checked"].get("text_ok"))
PY
done
for v in 1 2; do
  SBX_K1B_VARIANT=$v timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $OUT/bench_config2_k1b_variant$v.json 2> /tmp/bench_full_$v.err
  python - $OUT/bench_config2_k1b_variant$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], "Mreads/s", d["value"], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, "parity", d["parity_checked"]["ok"], d["parity_checked"].get("text_ok"))
PY
done
# SQ instruction counts of both variants on the 40-Mbp input (own PMC passes, no traces)
export TMPDIR=/tmp
REPO=$(pwd)
for v in 1 2; do
  ( cd /tmp && SBX_K1B_VARIANT=$v timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d /tmp/sq_v$v -o s -- \
      python $REPO/bench.py --length 40000000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> /tmp/sq_v$v.err )
  python - $v <<'PY' | tee -a $OUT/sq_insts_40Mbp.txt
import csv, glob, collections, re, sys
acc = collections.defaultdict(float)
for f in glob.glob("/tmp/sq_v%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", row["Kernel_Name"])
        acc[(m.group(1) if m else row["Kernel_Name"][:30], row["Counter_Name"])] += float(row["Counter_Value"])
for (k, c), v in sorted(acc.items()):
    if v > 1e6: print("variant", sys.argv[1], k, c, "%.4g" % v)
PY
done
