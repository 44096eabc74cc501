#!/bin/bash
# round 6, GPU call D: K1b with wide own-lane copies (two 8-byte words instead of four collapsing dwords) against the product body in
# tools/k1_lab, 40 Mbp and config 2's size; the sharded CLI tests again
set -u
OUT=gpurun_out/r6_d
mkdir -p $OUT
S=/dev/shm
timeout 120 tools/gen_bam --out $S/lab40.bam --contigs chr1:40000000 --coverage 30 --seed 0x5A4D0002 --level 6 --codec zlib > /dev/null 2> /tmp/gen40.err
timeout 300 tools/k1_lab $S/lab40.bam 5 ablate 2> $OUT/k1_lab_40Mbp.err | tee $OUT/k1_lab_40Mbp.jsonl
rm -f $S/lab40.bam $S/lab40.bam.bai
timeout 900 python -m pytest tests/test_gpu_cli_sharded.py -x -q -m gpu --durations=5 2>&1 | tail -15 | tee $OUT/tests_sharded.txt
timeout 300 tools/gen_bam --out $S/lab248.bam --contigs chr1:248956422 --coverage 30 --seed 0x5A4D0002 --level 6 --codec zlib > /dev/null 2> /tmp/gen248.err
timeout 300 tools/k1_lab $S/lab248.bam 3 2> $OUT/k1_lab_config2.err | tee $OUT/k1_lab_config2.jsonl
rm -f $S/lab248.bam $S/lab248.bam.bai
