#!/bin/bash
# round 4, call T: several BAMs with compatible but different @SQ dictionaries (merged like SamHeaderMerger; reference ids
# translated in K2, BAI queries on the host)
set -u
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_multibam.py -x -q 2>&1 | tail -8 | tee gpurun_out/t_multibam_tests.txt
