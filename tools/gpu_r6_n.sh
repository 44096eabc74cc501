#!/bin/bash
# round 6, GPU call N: K1's overlapped schedule (balanced K1a launch, its tail on a side stream next to K1b) -- the config-2 bench with it and
# with the one-launch schedule, the lab's sweep of waves per CU at full size, the inflate / depth tests
# (the two-stream schedule and the lab's `split` mode exist in commit 809ed8a only: the experiment was taken out again)
set -u
OUT=$(pwd)/gpurun_out/r6_n
mkdir -p $OUT
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], d["value"], "Mreads/s", d["ms_per_step"], "ms", {k: round(v["ms"], 2) for k, v in d["kernels"].items()}, "parity", d["parity_checked"].get("ok"), d["parity_checked"].get("coverage"))
except Exception as e:
    print("no line", e)
PY
}
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --parity-windows 8 --no-side-runs > $OUT/bench_overlap.json 2> $OUT/bench_overlap.err
echo "overlap rc=$?"; summ $OUT/bench_overlap.json
SBX_K1_WAVES_PER_CU=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --parity-windows 2 --no-full-parity --no-side-runs > $OUT/bench_one_launch.json 2> $OUT/bench_one_launch.err
echo "one launch rc=$?"; summ $OUT/bench_one_launch.json
SBX_LAB_CORUN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --parity-windows 2 --no-full-parity --no-side-runs > $OUT/bench_lab_corun_k2_next_to_k1b.json 2> $OUT/bench_lab_corun.err
echo "LAB K2 next to K1b rc=$?"; summ $OUT/bench_lab_corun_k2_next_to_k1b.json
BAM=$(ls -S /dev/shm/sbx_bench_*.bam | head -1)
echo "lab on $BAM"
timeout 900 tools/k1_lab $BAM 5 split 2> $OUT/k1_lab_split.err | tee $OUT/k1_lab_split.jsonl | grep -v '"check"'
tail -3 $OUT/k1_lab_split.err
timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_depth.py tests/test_gpu_batches.py -m gpu -x -q 2>&1 | tail -3
