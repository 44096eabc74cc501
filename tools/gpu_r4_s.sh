#!/bin/bash
# round 4, call S: sbx_build_index in batches of blocks (open-ended chain runs in K2) -- the new index tests, then the tests
# that lean on the record chain (repair, edge cases, work lists, depth fixtures).
set -u
mkdir -p gpurun_out
timeout 80 python -m pytest tests/test_gpu_writer.py -x -q -k "index" 2>&1 | tail -6 | tee gpurun_out/s_index_tests.txt
timeout 150 python -m pytest tests/test_gpu_repair.py tests/test_gpu_edge_cases.py tests/test_gpu_worklist.py tests/test_gpu_depth.py -x -q 2>&1 | tail -6 | tee gpurun_out/s_chain_tests.txt
