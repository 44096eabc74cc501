#!/bin/bash
# round 6, GPU call P: K6 with the register-assembled rows (format_core.hpp): the format / depth / pipeline tests, the config-2 line with its
# whole-share parity (8.25 GB of device text against the oracle) and `device_text`, a kernel trace of the text pass
set -u
OUT=$(pwd)/gpurun_out/r6_p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_format.py tests/test_gpu_depth.py tests/test_gpu_pipeline.py tests/test_gpu_compact.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_config2_k6.json 2> $OUT/bench_config2_k6.err
echo "bench rc=$?"
python - $OUT/bench_config2_k6.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(d["value"], "Mreads/s", d["ms_per_step"], "ms", {k: round(v["ms"], 2) for k, v in d["kernels"].items()}, "parity", d["parity_checked"].get("ok"), d["parity_checked"].get("coverage"))
    print("device_text", d.get("device_text"))
except Exception as e:
    print("no line", e)
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o k6 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 2 --no-full-parity > /dev/null 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
python tools/summarize_prof.py $OUT/prof 2>/dev/null | head -14 || find $OUT/prof -name "*kernel_stats*" | head
