#!/bin/bash
# round 4, call V: the BAI built on the device (bai_parallel.hpp, one lane per record) -- every index test again
set -u
mkdir -p gpurun_out
SBX_TIMING=1 timeout 100 python -m pytest tests/test_gpu_writer.py -x -q -k "index or unsorted or quirky" 2>&1 | grep -v "^\[sbx\] \(open\|bgzf\|hipMalloc\)" | tail -12 | tee gpurun_out/v_index_tests.txt
