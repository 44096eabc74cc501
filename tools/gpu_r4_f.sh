#!/bin/bash
# round 4, GPU call F: the shader clock under each kernel (GRBM_GUI_ACTIVE / duration), old and new K1a
OUT=$(pwd)/gpurun_out/r4f
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 0 > $OUT/warm.json 2> $OUT/warm.err
for k in 1 2; do
  SBX_K1A=$k rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/clk$k" -o s -- \
    python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> "$OUT/clk$k.err"
done
ls -R $OUT | head -30
python - <<PY
import csv, glob, collections
for k in (1, 2):
    cc = glob.glob("$OUT/clk%d/**/*counter_collection.csv" % k, recursive=True)
    kt = glob.glob("$OUT/clk%d/**/*kernel_trace.csv" % k, recursive=True)
    print("files", cc, kt)
    dur = {}
    for f in kt:
        for row in csv.DictReader(open(f)):
            dur[row["Dispatch_Id"]] = (row["Kernel_Name"], int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for f in cc:
        seen = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
            d = dur.get(row["Dispatch_Id"])
            if not d or d[1] < 1000000: continue
            seen[d[0][:40]].append((float(row["Counter_Value"]), d[1]))
        for name, v in seen.items():
            c = sum(x for x, _ in v) / len(v); t = sum(y for _, y in v) / len(v)
            print("K1A=%d" % k, name, "cycles %.4g  ns %.4g  -> %.0f MHz (sum over XCDs/SEs?)" % (c, t, c / t * 1e3))
PY
find $OUT -name '*counter_collection.csv' -size +2M -delete; find $OUT -name '*kernel_trace.csv' -size +2M -delete
