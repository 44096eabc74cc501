#!/bin/bash
# round 3, GPU call J2: wait states out of K1a's canonical decode and K1b's short copies
OUT=gpurun_out/r3j2
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_edge_cases.py tests/test_gpu_depth.py tests/test_gpu_repair.py -x -q > $OUT/t_default.log 2>&1; echo "default tests rc=$?"; tail -2 $OUT/t_default.log
SBX_K1B_VARIANT=0 SBX_K1A_BURST=4 timeout 400 python -m pytest tests/test_gpu_inflate.py -x -q > $OUT/t_old.log 2>&1; echo "old K1b / burst 4 tests rc=$?"; tail -1 $OUT/t_old.log
timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
echo "config 2 rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench_c2.json"))
print("config 2:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
PY
