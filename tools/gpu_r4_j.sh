#!/bin/bash
# round 4, GPU call J: nontemporal token stores in K1a2 (SBX_K1A_EXP=8) against the default, at 15 / 10 / 8 wavefronts per CU
OUT=gpurun_out/r4j
mkdir -p $OUT
SBX_K1A_EXP=8 timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q > $OUT/t.log 2>&1; echo "tests (nt) rc=$?"; tail -1 $OUT/t.log
for e in 0 8; do
 for pad in 0 5600 9700; do
  SBX_K1A_EXP=$e SBX_K1A_LDS_PAD=$pad timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 2 > $OUT/b${e}_${pad}.json 2> $OUT/b${e}_${pad}.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b${e}_${pad}.json"))
    print("exp $e pad $pad:", d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as ex:
    print("no line", ex); print(open("$OUT/b${e}_${pad}.err").read()[-800:])
PY
 done
done
