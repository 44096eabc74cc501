#!/bin/bash
# round 4, GPU call N: BASELINE configs 3, 4 (whole genome, zlib-6 input generated once for both) and 5 as measured lines: cpu_baseline, e2e,
# 32 parity samples, 5 steps, one FETCH_SIZE and one WRITE_SIZE pass each.  Raw profiler output stays in /tmp on the box.
OUT=$(pwd)/gpurun_out/r4n2
RAW=/tmp/r4n_raw
REPO=$(pwd)
mkdir -p $OUT $RAW
export TMPDIR=/tmp
for cfg in 3 4 5; do
  cd $REPO
  timeout 2400 python bench.py --config $cfg --steps 5 --warmup 1 --parity-windows 32 > $OUT/bench_config${cfg}_full.json 2> $RAW/bench_config$cfg.err
  echo "config $cfg rc=$?"; tail -c 400 $RAW/bench_config$cfg.err | tr '\n' ' '; echo
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_config${cfg}_full.json"))
    print("config $cfg:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, "parity", d["parity_checked"].get("ok"), d["parity_checked"].get("windows"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "e2e", (d.get("e2e") or {}).get("seconds"), d["host"])
except Exception as e:
    print("no line", e)
PY
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $RAW/pmc_c${cfg}_$c -o p -- python $REPO/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> $RAW/pmc_c${cfg}_$c.err
  done
  cd $REPO
  python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(float)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$RAW/pmc_c${cfg}_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+)", row["Kernel_Name"])
            acc[(m.group(1) if m else row["Kernel_Name"][:30], row["Counter_Name"])] += float(row["Counter_Value"])
with open("$OUT/pmc_fetch_write_config$cfg.csv", "w") as fh:
    fh.write("kernel,counter,value_KB,launch\n")
    for (k, c), v in sorted(acc.items()):
        if v >= 1024:
            fh.write("%s,%s,%d,0\n" % (k, c, v))
print("pmc config $cfg written:", len(acc))
PY
done
du -sh $OUT
