#!/bin/bash
# round 6, GPU call C: `sbx-depth --gpus N` (one process, N contexts) against one context and the oracle; the multibam refusal test again
set -u
OUT=gpurun_out/r6_c
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_cli_sharded.py tests/test_gpu_multibam.py -x -q -m gpu --durations=8 2>&1 | tail -40 | tee $OUT/tests.txt
