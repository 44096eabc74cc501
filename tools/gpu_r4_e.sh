#!/bin/bash
# round 4, GPU call E: SQ counters of huffman_decode2 at FULL size (3,379 wavefronts: the occupancy the kernel was built for)
OUT=$(pwd)/gpurun_out/r4e
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 0 > $OUT/warm.json 2> $OUT/warm.err   # generates the BAM once
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-include-regex "huffman_decode2" --output-format csv -d "$OUT/$name" -o s -- \
    python "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> "$OUT/$name.err"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
run sq3 SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_IFETCH
run sq4 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES_EQ_64 SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_WAVE32_LDS
python - <<PY
import csv, glob, collections
for d in ("sq1", "sq2", "sq3", "sq4"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    for k in sorted(acc): print(d, k, "%.5g" % acc[k], "dispatches", n[k])
PY
tail -2 $OUT/sq4.err
find $OUT -name '*counter_collection.csv' -size +2M -delete
