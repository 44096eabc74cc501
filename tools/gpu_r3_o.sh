#!/bin/bash
# round 3, GPU call O: kernel trace of config 5 (base -m -q20, 300x) at a quarter of its length -- where K7's time goes
OUT=gpurun_out/r3o
mkdir -p $OUT
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/kt -o kt -- \
    python $REPO/bench.py --config 5 --scale 0.25 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 2 > $REPO/$OUT/bench_c5.json 2> /dev/null; echo "rc=$?"
cd $REPO
find $OUT -name '*_kernel_trace.csv' -size +4M -delete
cut -c1-160 $OUT/kt/*kernel_stats.csv | head -30
