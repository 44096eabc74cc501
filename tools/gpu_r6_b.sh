#!/bin/bash
# round 6, GPU call B: the tests of the round's correctness changes (over-end alignments, -m with several files, window fallback, known answers)
set -u
OUT=gpurun_out/r6_b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_pileup_known_answers.py tests/test_gpu_edge_cases.py tests/test_gpu_multibam.py tests/test_gpu_windows.py tests/test_gpu_compact.py -x -q -m gpu --durations=8 2>&1 | tail -25 | tee $OUT/tests.txt
