#!/bin/bash
# round 3, GPU call Q: config 5 at full size on the final code
OUT=gpurun_out/r3q
mkdir -p $OUT
timeout 400 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 6 > $OUT/bench_config5.json 2> /dev/null; echo "config 5 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3q/bench_config5.json"))
print("config 5:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
PY
