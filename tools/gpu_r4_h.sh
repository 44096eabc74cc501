#!/bin/bash
# round 4, GPU call H: what the token stores cost K1a2 (SBX_K1A_EXP: 1 = stores to a fixed address, 2 = no stores; the general kernel re-decodes, +27.7 ms)
OUT=gpurun_out/r4h
mkdir -p $OUT
for e in 0 1 2; do
 for pad in 0 9700; do
  SBX_K1A_EXP=$e SBX_K1A_LDS_PAD=$pad timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 2 > $OUT/b_$e_$pad.json 2> $OUT/b_$e_$pad.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$e_$pad.json"))
    print("exp $e pad $pad:", d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as ex:
    print("no line", ex); print(open("$OUT/b_$e_$pad.err").read()[-800:])
PY
 done
done
