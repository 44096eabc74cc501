#!/bin/bash
# tools/k1_lab on the 40-Mbp bench-like BAM (a quick A/B of inflate kernel variants); $1 = output tag, $2... = extra k1_lab arguments
set -u
TAG=${1:-lab}
shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
S=/dev/shm
timeout 120 tools/gen_bam --out $S/lab40.bam --contigs chr1:40000000 --coverage 30 --seed 0x5A4D0002 --level 6 --codec zlib > /dev/null 2> /tmp/gen40.err
timeout 400 tools/k1_lab $S/lab40.bam 5 "$@" 2> $OUT/k1_lab_40Mbp.err | tee $OUT/k1_lab_40Mbp.jsonl
tail -3 $OUT/k1_lab_40Mbp.err
