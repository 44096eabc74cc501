#!/bin/bash
# round 3, final GPU call: the measurement set of profiles/round3 (bench line as the driver runs it, kernel trace, PMC passes),
# the whole GPU test-suite with its slowest tests, config 5 at full size
OUT=gpurun_out/r3k
mkdir -p $OUT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_invocation.json 2> $OUT/bench_driver_invocation.err; echo "bench (driver's invocation) rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3k/bench_driver_invocation.json"))
print(d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"], d["roofline"]["frac"], d["cpu_baseline"], {k:d["e2e"].get(k) for k in ("seconds","Mreads_per_s","single_process_seconds","all_seconds")})
PY
timeout 1500 bash tools/profile_round.sh $OUT/prof 3 > $OUT/profile_round.log 2>&1; echo "profile_round rc=$?"; tail -3 $OUT/profile_round.log
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -32 $OUT/gpu_tests.log
timeout 1200 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 6 > $OUT/bench_config5.json 2> /dev/null; echo "config 5 rc=$?"
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r3k/bench_config5.json"))
    print("config 5:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as e: print("no config 5 line", e)
PY
timeout 900 python bench.py --config 4 --scale 0.15 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 6 > $OUT/bench_config4_scale015.json 2> $OUT/bench_config4.err; echo "config 4 (scale 0.15) rc=$?"
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r3k/bench_config4_scale015.json"))
    print("config 4 (scale 0.15):", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"], d["host"])
except Exception as e: print("no config 4 line", e)
PY
find $OUT -name '*.csv' -size +6M -delete
du -sh $OUT
