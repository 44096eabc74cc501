#!/bin/bash
# round 3, GPU call P: find_mates scanning four window entries per step
OUT=gpurun_out/r3p
mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_mates.py tests/test_gpu_random_differential.py tests/test_gpu_region_window.py -x -q > $OUT/t.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/t.log
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/kt -o kt -- \
    python $REPO/bench.py --config 5 --scale 0.25 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 4 > $REPO/$OUT/bench_c5.json 2> /dev/null; echo "rc=$?"
cd $REPO
find $OUT -name '*_kernel_trace.csv' -size +4M -delete
python - <<'PY'
import csv, re, json
for r in csv.DictReader(open('gpurun_out/r3p/kt/kt_kernel_stats.csv')):
    m = re.search(r'(k_[a-z0-9_]+)', r['Name'])
    if m and float(r['AverageNs']) > 2e5: print(m.group(1), r['Calls'], round(float(r['AverageNs']) / 1e6, 3), 'ms avg')
d = json.load(open('gpurun_out/r3p/bench_c5.json'))
print("config 5 (scale 0.25):", d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, d["parity_checked"]["ok"])
PY
