#!/bin/bash
# round 6, GPU call Q: kernel trace of the config-2 bench with its device_text side run (what K6's two kernels cost per piece now)
set -u
OUT=$(pwd)/gpurun_out/r6_q
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 0 --no-full-parity > $OUT/bench_under_rocprofv3.json 2> $OUT/kt.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats_config2_device_text.csv
python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf $OUT/kt
