#!/bin/bash
# round 6, GPU call O: K1's overlapped schedule with the balanced launch in workgroups of FOUR wavefronts (one per SIMD): the lab's sweep at full
# size, the config-2 bench with it
# (the two-stream schedule and the lab's `split` mode exist in commit 809ed8a only: the experiment was taken out again)
set -u
OUT=$(pwd)/gpurun_out/r6_o
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
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --parity-windows 8 --no-side-runs > $OUT/bench_overlap_wg4.json 2> $OUT/bench_overlap_wg4.err
echo "overlap rc=$?"; summ $OUT/bench_overlap_wg4.json
BAM=$(ls -S /dev/shm/sbx_bench_*.bam | head -1)
echo "lab on $BAM"
timeout 900 tools/k1_lab $BAM 5 split 2> $OUT/k1_lab_split_wg4.err | tee $OUT/k1_lab_split_wg4.jsonl | grep -v '"check"\|"k1b"'
tail -3 $OUT/k1_lab_split_wg4.err
timeout 600 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q 2>&1 | tail -3
