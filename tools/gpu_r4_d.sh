#!/bin/bash
# round 4, GPU call D: K1a after the instruction diet -- parity, then config 2
OUT=gpurun_out/r4d
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $OUT/t_inflate.log 2>&1; echo "inflate tests rc=$?"; tail -2 $OUT/t_inflate.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json"))
    print(d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as e:
    print("no line", e); print(open("$OUT/bench.err").read()[-1500:])
PY
