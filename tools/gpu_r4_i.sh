#!/bin/bash
# round 4, GPU call I: which part of the token flush costs K1a2 its 7 ms (SBX_K1A_EXP bits: 2 = no store instruction, 4 = no staging reads;
# with any bit set the general kernel re-decodes everything, + ~27.7 ms)
OUT=gpurun_out/r4i
mkdir -p $OUT
for e in 0 1 2 4 6; do
  SBX_K1A_EXP=$e timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 2 > $OUT/b$e.json 2> $OUT/b$e.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b$e.json"))
    print("exp $e:", d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as ex:
    print("no line", ex); print(open("$OUT/b$e.err").read()[-800:])
PY
done
