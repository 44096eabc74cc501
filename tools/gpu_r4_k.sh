#!/bin/bash
# round 4, GPU call K: the overlapped inflate schedule (K1a of the second part of the blocks next to K1b of the first): split point sweep
OUT=gpurun_out/r4k
mkdir -p $OUT
SBX_INFLATE_SPLIT=50 timeout 300 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_large_properties.py -x -q > $OUT/t.log 2>&1; echo "tests rc=$?"; tail -1 $OUT/t.log
for sp in 0 40 50 60 70; do
 for pad in 0 5600; do
  SBX_INFLATE_SPLIT=$sp SBX_K1A_LDS_PAD=$pad timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/b${sp}_${pad}.json 2> $OUT/b${sp}_${pad}.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b${sp}_${pad}.json"))
    k={k:v["ms"] for k,v in d["kernels"].items()}
    print("split $sp pad $pad:", d["ms_per_step"], "inflate", round(k["huffman_decode"]+k["lz77_resolve"],2), k, d["parity_checked"]["ok"])
except Exception as ex:
    print("no line", ex); print(open("$OUT/b${sp}_${pad}.err").read()[-800:])
PY
 done
done
