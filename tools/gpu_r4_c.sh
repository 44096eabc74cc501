#!/bin/bash
# round 4, GPU call C: where the new K1a's time goes -- kernel trace at full size, SQ counters of huffman_decode2 on the 90 Mbp sub-problem
OUT=$(pwd)/gpurun_out/r4c
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- \
    python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
python - <<PY
import csv, glob
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print(row["Name"][:60], row["Calls"], row["AverageNs"], row["Percentage"])
PY
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-include-regex "huffman|translate" --output-format csv -d "$OUT/sq1" -o s -- \
    python "$REPO/bench.py" --length 90000000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> "$OUT/sq1.err"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU --kernel-include-regex "huffman|translate" --output-format csv -d "$OUT/sq2" -o s -- \
    python "$REPO/bench.py" --length 90000000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> "$OUT/sq2.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --kernel-include-regex "huffman|translate" --output-format csv -d "$OUT/sq3" -o s -- \
    python "$REPO/bench.py" --length 90000000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> "$OUT/sq3.err"
python - <<PY
import csv, glob, collections
for d in ("sq1", "sq2", "sq3"):
    acc = collections.defaultdict(float)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"][:50], row["Counter_Name"])] += float(row["Counter_Value"])
    for k in sorted(acc): print(d, k[0], k[1], "%.4g" % acc[k])
PY
tail -3 $OUT/sq3.err
find $OUT -name '*_kernel_trace.csv' -size +8M -delete
find $OUT -name '*counter_collection.csv' -size +2M -delete
