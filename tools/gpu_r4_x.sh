#!/bin/bash
# round 4, call X (the last seconds of the budget): single-file filter / region / window tests after the reference-id translation
# went into K2's describe -- as many as fit 20 s.
set -u
mkdir -p gpurun_out
timeout 20 python -m pytest tests/test_gpu_filters.py tests/test_gpu_region_window.py tests/test_gpu_batches.py -x -q 2>&1 | tail -5 | tee gpurun_out/x_tests.txt
