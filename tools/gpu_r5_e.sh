#!/bin/bash
# round 5, GPU call E: the tests call D stopped in front of, the multi-workgroup scans of K2 (tests, then A/B at full size)
set -u
OUT=gpurun_out/r5_e
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_region_window.py tests/test_gpu_windows.py tests/test_gpu_batches.py tests/test_gpu_depth.py tests/test_gpu_repair.py tests/test_gpu_edge_cases.py tests/test_gpu_worklist.py tests/test_gpu_writer.py -q --durations=5 2>&1 | tail -15 | tee $OUT/tests_a.txt
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_dist.py -q --durations=5 2>&1 | tail -15 | tee $OUT/tests_b.txt
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
for m in 0 1; do
  SBX_K2_SCAN=$m timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-full-parity > $OUT/bench_config2_scan$m.json 2> /tmp/bs_$m.err; show $OUT/bench_config2_scan$m.json
done
