#!/bin/bash
# round 3, GPU call I: K1b with / without the two compiler hints, K7 worklist, index of files whose header fills its blocks
OUT=gpurun_out/r3i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_writer.py tests/test_gpu_mates.py tests/test_gpu_random_differential.py tests/test_gpu_inflate.py -x -q > $OUT/t_default.log 2>&1; echo "default tests rc=$?"; tail -2 $OUT/t_default.log
export SBX_TIMING=1
for v in 1 4 5 6 7 3; do
  SBX_K1B_VARIANT=$v timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_k1b$v.json 2> $OUT/bench_k1b$v.err
  echo "K1b $v rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_k1b$v.json"))
    print("K1b variant $v:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as e: print("no line", e)
PY
done
for v in 6 4; do SBX_K1B_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_edge_cases.py -x -q > $OUT/t_k1b$v.log 2>&1; echo "K1b variant $v tests rc=$?"; tail -1 $OUT/t_k1b$v.log; done
# config 5 at a quarter of its length: K7 with the worklist (the per-read cost is what is compared: 71.1 ms for 100 M reads before)
for k7 in 1 0; do
  SBX_K7_VARIANT=$k7 timeout 900 python bench.py --config 5 --scale 0.25 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_c5_k7$k7.json 2> $OUT/bench_c5_k7$k7.err
  echo "config 5 K7 variant $k7 rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c5_k7$k7.json"))
    print("config 5 (scale 0.25) K7 variant $k7:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d.get("mates"), d["parity_checked"]["ok"])
except Exception as e: print("no line", e)
PY
  grep -i "mates\|accumulate" $OUT/bench_c5_k7$k7.err | tail -3
done
