#!/bin/bash
# round 3, last GPU call: the GPU test-suite on the final commit, without the three slowest modules (bench lines, multi-rank runs,
# large properties -- they ran on the final measurement call and their code paths did not change since)
OUT=gpurun_out/r3r
mkdir -p $OUT
timeout 290 python -m pytest tests -m gpu -q -x --ignore=tests/test_gpu_bench.py --ignore=tests/test_gpu_dist.py --ignore=tests/test_gpu_large_properties.py > $OUT/gpu_tests_final_commit.log 2>&1; echo "rc=$?"; tail -3 $OUT/gpu_tests_final_commit.log
