#!/bin/bash
# round 6, GPU call K: describe with in-line piece staging: 40-Mbp bench (stderr kept), tests, config 2
set -u
OUT=gpurun_out/r6_k
mkdir -p $OUT
show() {
python - $1 <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], "Mreads/s", d["value"], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, "parity", d["parity_checked"]["ok"], (d["parity_checked"].get("full_text") or {}).get("ok"))
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
timeout 300 python bench.py --length 40000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 --no-side-runs > $OUT/bench_40Mbp.json 2> $OUT/bench_40Mbp.err
echo "rc=$?"; tail -3 $OUT/bench_40Mbp.err | cut -c1-300
show $OUT/bench_40Mbp.json
timeout 900 python -m pytest tests/test_gpu_depth.py tests/test_gpu_edge_cases.py tests/test_gpu_filters.py tests/test_gpu_repair.py tests/test_gpu_inflate.py tests/test_gpu_large_properties.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/tests.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-side-runs > $OUT/bench_config2.json 2> $OUT/bench_config2.err
echo "rc=$?"; tail -3 $OUT/bench_config2.err | cut -c1-300
show $OUT/bench_config2.json
