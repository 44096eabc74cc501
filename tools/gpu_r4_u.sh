#!/bin/bash
# round 4, call U: regression check after the K2 changes (open-ended runs, reference id translation): the writer tests that had
# not run on the device since the encoder became level-aware, then the 40-Mbp line of call O again (K2 was 1.90 ms there).
set -u
mkdir -p gpurun_out
timeout 70 python -m pytest tests/test_gpu_writer.py -x -q -k "many_blocks or unsorted or at_scale or device_codec" 2>&1 | tail -4 | tee gpurun_out/u_writer_rest.txt
timeout 60 python bench.py --length 40000000 --level 1 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 12 > gpurun_out/u_bench_40Mbp_level1.json 2> gpurun_out/u_bench.err
tail -c 1500 gpurun_out/u_bench_40Mbp_level1.json
