#!/bin/bash
# round 4, GPU call O: the new K1a on other compressors' streams at scale -- config 2 at 1/6 length, zlib levels 1 and 9, parity against the oracle
OUT=gpurun_out/r4o
mkdir -p $OUT
for lv in 1 9; do
  timeout 600 python bench.py --length 40000000 --level $lv --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 12 > $OUT/bench_level$lv.json 2> $OUT/bench_level$lv.err
  echo "level $lv rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_level$lv.json"))
    print("level $lv:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"], d["parity_checked"]["windows"], d["config"]["workload"][-60:])
except Exception as e:
    print("no line", e); print(open("$OUT/bench_level$lv.err").read()[-1000:])
PY
done
