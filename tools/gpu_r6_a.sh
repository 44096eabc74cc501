#!/bin/bash
# round 6, GPU call A: where K1b's time goes -- the batch loop with parts compiled out (tools/k1_lab.hip; the ablated launches produce
# invalid bytes on purpose), at 40 Mbp and at config 2's full size; then the driver's invocation as this round's starting line.
set -u
OUT=gpurun_out/r6_a
mkdir -p $OUT
export TMPDIR=/tmp
S=/dev/shm
timeout 120 tools/gen_bam --out $S/lab40.bam --contigs chr1:40000000 --coverage 30 --seed 0x5A4D0002 --level 6 --codec zlib > /dev/null 2> /tmp/gen40.err
timeout 200 tools/k1_lab $S/lab40.bam 5 2> $OUT/k1_lab_40Mbp.err | tee $OUT/k1_lab_40Mbp.jsonl
rm -f $S/lab40.bam $S/lab40.bam.bai
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_config2_driver_invocation.json 2> $OUT/bench_config2_driver_invocation.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_a/bench_config2_driver_invocation.json"))
print("config2", d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, d["parity_checked"]["ok"])
PY
B=$(ls $S/sbx_bench_*.bam | head -1)
timeout 300 tools/k1_lab $B 3 2> $OUT/k1_lab_config2.err | tee $OUT/k1_lab_config2.jsonl
