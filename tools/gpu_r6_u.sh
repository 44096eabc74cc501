#!/bin/bash
# round 6, GPU call U: K2 with the entry guess as a wave-per-block kernel and the describe kernel built around the simple -F evaluator:
# the tests that lean on the record chain and the filters, the config-2 bench (A/B: SBX_K2_SIMPLE_FILTER=0), a kernel trace
set -u
OUT=$(pwd)/gpurun_out/r6_u
mkdir -p $OUT
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], d["value"], "Mreads/s", d["ms_per_step"], "ms", {k: round(v["ms"], 2) for k, v in d["kernels"].items()}, "parity", d["parity_checked"].get("ok"), d["parity_checked"].get("coverage"))
except Exception as e:
    print("no line", e)
PY
}
timeout 1200 python -m pytest tests/test_gpu_repair.py tests/test_gpu_edge_cases.py tests/test_gpu_filters.py tests/test_gpu_worklist.py tests/test_gpu_depth.py tests/test_gpu_random_differential.py tests/test_gpu_batches.py tests/test_gpu_writer.py tests/test_gpu_region_window.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --parity-windows 8 --no-side-runs > $OUT/bench_config2_k2.json 2> $OUT/bench_config2_k2.err
echo "rc=$?"; summ $OUT/bench_config2_k2.json
SBX_K2_SIMPLE_FILTER=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --parity-windows 2 --no-full-parity --no-side-runs > $OUT/bench_config2_k2_interpreter.json 2> $OUT/bench_config2_k2_interpreter.err
echo "interpreter rc=$?"; summ $OUT/bench_config2_k2_interpreter.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 0 --no-full-parity --no-side-runs > $OUT/bench_under_rocprofv3.json 2> $OUT/kt.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats_config2.csv
python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MaxNs"], r["Percentage"])
PY
rm -rf $OUT/kt
