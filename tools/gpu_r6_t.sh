#!/bin/bash
# round 6, GPU call T: the rate of scattered 128-byte lines (tools/calib5: the roof of K2 and of K3's sequence fetches) at two footprints;
# the new text test (eight samples: rows past the LDS)
set -u
OUT=$(pwd)/gpurun_out/r6_t
mkdir -p $OUT
timeout 300 tools/calib5 14 | tee $OUT/calib5_scattered_lines_14GB.jsonl
timeout 300 tools/calib5 2 | tee $OUT/calib5_scattered_lines_2GB.jsonl
timeout 600 python -m pytest tests/test_gpu_format.py -m gpu -x -q 2>&1 | tail -3
