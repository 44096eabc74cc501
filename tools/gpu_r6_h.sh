#!/bin/bash
# round 6, GPU call H: after pruning the A/B variants and trimming the test matrices: the lab (zlib check + ablation of the product body),
# then the whole GPU suite with durations
set -u
OUT=gpurun_out/r6_h
mkdir -p $OUT
bash tools/gpu_lab40.sh r6_h ablate | grep -v '"check"'
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 > $OUT/gpu_tests.log 2>&1
tail -28 $OUT/gpu_tests.log
