#!/bin/bash
# round 3, GPU call J: K1b default (own-lane 32, 8 waves), K7 span counts of the fast path through a difference array
OUT=gpurun_out/r3j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mates.py tests/test_gpu_random_differential.py tests/test_gpu_inflate.py tests/test_gpu_edge_cases.py -x -q > $OUT/t_default.log 2>&1; echo "default tests rc=$?"; tail -2 $OUT/t_default.log
export SBX_TIMING=1
timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
echo "config 2 rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench_c2.json"))
print("config 2:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
PY
for k7 in 1; do
  SBX_K7_VARIANT=$k7 timeout 900 python bench.py --config 5 --scale 0.25 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_c5_k7$k7.json 2> /dev/null
  echo "config 5 K7 variant $k7 rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c5_k7$k7.json"))
    print("config 5 (scale 0.25) K7 variant $k7:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as e: print("no line", e)
PY
done
