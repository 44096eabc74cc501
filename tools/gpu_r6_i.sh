#!/bin/bash
# round 6, GPU call I: `describe` streams the records through LDS pieces (coalesced loads) -- tests, then kernel times at 40 Mbp and config 2
set -u
OUT=gpurun_out/r6_i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_depth.py tests/test_gpu_edge_cases.py tests/test_gpu_filters.py tests/test_gpu_repair.py tests/test_gpu_worklist.py tests/test_gpu_mates.py tests/test_gpu_multibam.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/tests.txt
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
timeout 200 python bench.py --length 40000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_40Mbp.json 2> /tmp/b40.err
show $OUT/bench_40Mbp.json
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e > $OUT/bench_config2.json 2> /tmp/bfull.err
show $OUT/bench_config2.json
