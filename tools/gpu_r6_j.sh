#!/bin/bash
# round 6, GPU call J: what aborted in call I's bench runs -- the lab's zlib check of K1b at 40 Mbp, then the 40-Mbp bench with its stderr kept
set -u
OUT=gpurun_out/r6_j
mkdir -p $OUT
bash tools/gpu_lab40.sh r6_j | grep -v '"check"' | head -8
timeout 300 python bench.py --length 40000000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 4 --no-side-runs > $OUT/bench_40Mbp.json 2> $OUT/bench_40Mbp.err
echo "rc=$?"; tail -15 $OUT/bench_40Mbp.err | cut -c1-400
