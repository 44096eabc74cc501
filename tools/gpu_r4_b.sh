#!/bin/bash
# round 4, GPU call B: the new K1a (huffman_decode2 + literal translation) -- parity tests, then A/B against round 3's kernel on config 2
OUT=gpurun_out/r4b
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $OUT/t_inflate.log 2>&1; echo "inflate tests rc=$?"; tail -3 $OUT/t_inflate.log
for k in 1 2; do
  SBX_K1A=$k timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_k1a$k.json 2> $OUT/bench_k1a$k.err
  echo "K1A=$k rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_k1a$k.json"))
    print("K1A=$k:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as e:
    print("no line", e); print(open("$OUT/bench_k1a$k.err").read()[-1500:])
PY
done
