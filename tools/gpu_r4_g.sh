#!/bin/bash
# round 4, GPU call G: K1a2 against occupancy (LDS padding: 15, 12, 10, 8, 6 wavefronts per CU)
OUT=gpurun_out/r4g
mkdir -p $OUT
for pad in 0 2900 5600 9700 16500; do
  SBX_K1A_LDS_PAD=$pad timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 0 > $OUT/bench_pad$pad.json 2> $OUT/bench_pad$pad.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_pad$pad.json"))
    print("pad $pad waves/CU", 163840 // (10688 + $pad), d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()})
except Exception as e:
    print("no line", e); print(open("$OUT/bench_pad$pad.err").read()[-800:])
PY
done
