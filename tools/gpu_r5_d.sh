#!/bin/bash
# round 5, GPU call D: compact region / window counters, the whole-contig parity of bench.py at development scale, the --gpus 2 line
set -u
OUT=gpurun_out/r5_d
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_compact.py tests/test_gpu_region_window.py tests/test_gpu_windows.py tests/test_gpu_batches.py tests/test_gpu_bench.py -x -q --durations=8 2>&1 | tail -25 | tee $OUT/tests_compact_bench.txt
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_mates.py tests/test_gpu_multibam.py -x -q 2>&1 | tail -6 | tee $OUT/tests_dist_mates_multibam.txt
# config 3 / 4 at 2 % of the genome: compact against full counters
for c in 1 0; do
  SBX_COMPACT=$c timeout 300 python bench.py --config 3 --scale 0.02 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 8 > $OUT/bench_config3_scale002_compact$c.json 2> /tmp/c3_$c.err
  python - $OUT/bench_config3_scale002_compact$c.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, "sum", round(sum(v["ms"] for v in d["kernels"].values()), 2), d["parity_checked"]["ok"], d["parity_checked"].get("whole_contig"))
except Exception as e:
    print("no line", e)
PY
done
tail -3 /tmp/c3_1.err
