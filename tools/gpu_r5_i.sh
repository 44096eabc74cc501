#!/bin/bash
# round 5, GPU call I (the last minutes of the budget): the work list / selection table kept between identical runs (config 4: 66 ms of
# host work per pass) and the multi-workgroup scans without their grid cap (config 3: 757 workgroups of tiles) -- a few tests, then both
# configs at full scale, kernel passes only.
OUT=$(pwd)/gpurun_out/r5_i
mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_worklist.py tests/test_gpu_region_window.py tests/test_gpu_batches.py -q -x 2>&1 | tail -3 | tee $OUT/tests.txt
for cfg in 4 3; do
  SBX_TIMING=1 timeout 760 python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-full-parity --parity-windows 8 > $OUT/bench_config${cfg}_full_kernels_only.json 2> /tmp/b$cfg.err
  echo "config $cfg rc=$?"; grep "run: work list" /tmp/b$cfg.err | tail -3
  python - $OUT/bench_config${cfg}_full_kernels_only.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(d["value"], "Mreads/s", d["ms_per_step"], "ms", {k: v["ms"] for k, v in d["kernels"].items()}, "sum", round(sum(v["ms"] for v in d["kernels"].values()), 1), "parity", d["parity_checked"]["ok"], d["parity_checked"]["windows"])
except Exception as e:
    print("no line", e)
PY
done
