#!/bin/bash
# round 3, GPU call M: smoke() of the driver's entry point, then BASELINE configs 3 and 4 at FULL scale on the final code
OUT=gpurun_out/r3m
mkdir -p $OUT
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 900 python bench.py --config 3 --scale 1.0 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 18 > $OUT/bench_config3_full.json 2> $OUT/bench_config3.err; echo "config 3 rc=$?"
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r3m/bench_config3_full.json"))
    print("config 3 full:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"], d["host"])
except Exception as e: print("no config 3 line", e)
PY
timeout 300 python bench.py --config 4 --scale 1.0 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 18 > $OUT/bench_config4_full.json 2> $OUT/bench_config4.err; echo "config 4 rc=$?"
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r3m/bench_config4_full.json"))
    print("config 4 full:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"], d["host"])
except Exception as e: print("no config 4 line", e)
PY
tail -c 600 $OUT/bench_config3.err
