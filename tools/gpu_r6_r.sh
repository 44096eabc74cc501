#!/bin/bash
# round 6, GPU call R: K6 two-pass with the coalesced count scan and LDS sized from the row length: the tests of every text path, the config-2 line with
# whole-share parity and device_text, kernel stats of the text pass
set -u
OUT=$(pwd)/gpurun_out/r6_r
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_format.py tests/test_gpu_depth.py tests/test_gpu_pipeline.py tests/test_gpu_cli_sharded.py tests/test_gpu_multibam.py tests/test_gpu_batches.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --parity-windows 4 > $OUT/bench_config2_k6_prepared_rows.json 2> $OUT/bench_config2_k6_prepared_rows.err
echo "bench rc=$?"
python - $OUT/bench_config2_k6_prepared_rows.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(d["value"], "Mreads/s", d["ms_per_step"], "ms", {k: round(v["ms"], 2) for k, v in d["kernels"].items()}, "parity", d["parity_checked"].get("ok"), d["parity_checked"].get("coverage"))
    print("device_text", d.get("device_text"))
    print("e2e", d.get("e2e"))
except Exception as e:
    print("no line", e)
PY
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 0 --no-full-parity > $OUT/bench_under_rocprofv3.json 2> $OUT/kt.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats_config2_device_text.csv
python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf $OUT/kt
