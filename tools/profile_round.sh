#!/bin/bash
# Collects the measurement set kept under profiles/<round>/ on a GPU box (run from the repo root):
#   bench_n1.json              python bench.py (full workload, with the CPU baseline)
#   kt/                        rocprofv3 --kernel-trace --stats of the same bench
#   fetch/ write/              PMC passes FETCH_SIZE / WRITE_SIZE (own runs: counters are never combined with traces)
#   sq1/ sq2/                  SQ occupancy / issue counters (full size: the wavefront count decides the occupancy of the lane-per-block kernel)
# Usage: tools/profile_round.sh <outdir under gpurun_out> [steps] [extra bench.py arguments, e.g. "--config 3 --codec device"]
set -u
OUT=${1:-gpurun_out/prof}
STEPS=${2:-3}
EXTRA=${3:-}
REPO=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
python bench.py $EXTRA --steps "$STEPS" --warmup 1 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- \
    python "$REPO/bench.py" $EXTRA --steps "$STEPS" --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- \
    python "$REPO/bench.py" $EXTRA --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- \
    python "$REPO/bench.py" $EXTRA --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq1" -o s -- \
    python "$REPO/bench.py" $EXTRA --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> "$OUT/sq1.err"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d "$OUT/sq2" -o s -- \
    python "$REPO/bench.py" $EXTRA --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> "$OUT/sq2.err"
cd "$REPO"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
# raw traces are large: keep the summaries
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
tail -5 "$OUT/summary.txt"
