#!/bin/bash
# round 6, GPU call G: the whole GPU suite on the round's code so far (with durations: which parametrisations to trim), then the driver's line
set -u
OUT=gpurun_out/r6_g
mkdir -p $OUT
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=40 > $OUT/gpu_tests.log 2>&1
tail -60 $OUT/gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_config2_driver_invocation.json 2> $OUT/bench_config2_driver_invocation.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_g/bench_config2_driver_invocation.json"))
print("config2", d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, d["parity_checked"]["ok"], d.get("device_text"))
PY
