#!/bin/bash
# round 4, GPU call L: BASELINE configs 3, 4 (whole genome, one generated BAM for both; --codec device for the generation) and 5 as measured
# lines: cpu_baseline, e2e, >= 32 parity samples, 5 steps, one FETCH_SIZE and one WRITE_SIZE pass each, a kernel trace of config 3
OUT=$(pwd)/gpurun_out/r4l
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
for cfg in 3 4 5; do
  codec="--codec device"
  timeout 1500 python bench.py --config $cfg $codec --steps 5 --warmup 1 --parity-windows 32 > $OUT/bench_config$cfg.json 2> $OUT/bench_config$cfg.err
  echo "config $cfg rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_config$cfg.json"))
    print("config $cfg:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, "parity", d["parity_checked"].get("ok"), d["parity_checked"].get("windows"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "e2e", (d.get("e2e") or {}).get("seconds"), d["host"])
except Exception as e:
    print("no line", e); print(open("$OUT/bench_config$cfg.err").read()[-1500:])
PY
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_c${cfg}_$c -o p -- python $REPO/bench.py --config $cfg $codec --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> $OUT/pmc_c${cfg}_$c.err
  done
  if [ $cfg = 3 ]; then
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c3 -o kt -- python $REPO/bench.py --config 3 $codec --steps 2 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> $OUT/kt_c3.err
    python - <<PY
import csv, glob
for f in glob.glob("$OUT/kt_c3/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f)))[:16]:
        print(row["Name"][:70], row["Calls"], row["TotalDurationNs"], row["Percentage"])
PY
  fi
  cd $REPO
  python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(float)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/pmc_c${cfg}_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+)", row["Kernel_Name"])
            acc[(m.group(1) if m else row["Kernel_Name"][:30], row["Counter_Name"])] += float(row["Counter_Value"])
with open("$OUT/pmc_fetch_write_config$cfg.csv", "w") as fh:
    fh.write("kernel,counter,value_KB,launch\n")
    for (k, c), v in sorted(acc.items()):
        if v >= 1024:
            fh.write("%s,%s,%d,0\n" % (k, c, v)); print("pmc config $cfg", k, c, "%.2f GB" % (v * 1024 / 1e9))
PY
  find $OUT -name '*counter_collection.csv' -size +1M -delete; find $OUT -name '*kernel_trace.csv' -size +1M -delete
done
