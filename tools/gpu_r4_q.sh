#!/bin/bash
# round 4, GPU call Q: find_mates as a hash join -- parity (mates tests, the random differential), then config 5 at a quarter of its length, scan against join
OUT=gpurun_out/r4q
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mates.py tests/test_gpu_random_differential.py -x -q > $OUT/t.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/t.log
for f in 0 1; do
  SBX_K7_FIND=$f timeout 600 python bench.py --config 5 --scale 0.25 --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 8 > $OUT/bench_c5_find$f.json 2> $OUT/bench_c5_find$f.err
  echo "find $f rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c5_find$f.json"))
    print("find $f:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as e:
    print("no line", e); print(open("$OUT/bench_c5_find$f.err").read()[-800:])
PY
done
