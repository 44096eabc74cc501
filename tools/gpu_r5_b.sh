#!/bin/bash
# round 5, GPU call B: K1b variant 3 (exact readiness rule) and the staged describe kernel -- correctness, then A/B.
set -u
OUT=gpurun_out/r5_b
mkdir -p $OUT
SBX_K1B_VARIANT=3 timeout 240 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_depth.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -6 | tee $OUT/tests_k1b_variant3.txt
timeout 500 python -m pytest tests/test_gpu_depth.py tests/test_gpu_filters.py tests/test_gpu_edge_cases.py tests/test_gpu_repair.py tests/test_gpu_worklist.py tests/test_gpu_multibam.py tests/test_gpu_writer.py -x -q 2>&1 | tail -6 | tee $OUT/tests_describe_staged.txt
SBX_K2_DESCRIBE=2 timeout 200 python -m pytest tests/test_gpu_depth.py tests/test_gpu_filters.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -4 | tee $OUT/tests_describe_staged_w4.txt
show() {
python - $1 <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], "Mreads/s", d["value"], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, "parity", d["parity_checked"]["ok"], d["parity_checked"].get("text_ok"), d["parity_checked"].get("full_text"))
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
for v in 1 3; do for f in 0 1 2; do
  SBX_K1B_VARIANT=$v SBX_K2_DESCRIBE=$f timeout 120 python bench.py --length 40000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 --no-full-parity > $OUT/bench_40Mbp_k1b${v}_desc$f.json 2> /tmp/b40_${v}_$f.err
  show $OUT/bench_40Mbp_k1b${v}_desc$f.json
done; done
SBX_K1B_VARIANT=1 SBX_K2_DESCRIBE=0 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-full-parity > $OUT/bench_config2_k1b1_desc0.json 2> /tmp/bf_1_0.err; show $OUT/bench_config2_k1b1_desc0.json
SBX_K1B_VARIANT=3 SBX_K2_DESCRIBE=1 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $OUT/bench_config2_k1b3_desc1.json 2> /tmp/bf_3_1.err; show $OUT/bench_config2_k1b3_desc1.json; tail -5 /tmp/bf_3_1.err
SBX_K1B_VARIANT=3 SBX_K2_DESCRIBE=2 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-full-parity > $OUT/bench_config2_k1b3_desc2.json 2> /tmp/bf_3_2.err; show $OUT/bench_config2_k1b3_desc2.json
