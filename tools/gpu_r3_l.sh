#!/bin/bash
# round 3, GPU call L: token stores of K1a as nontemporal stores -- time, FETCH_SIZE and WRITE_SIZE per variant
OUT=gpurun_out/r3l
mkdir -p $OUT
for b in 21 22; do
  SBX_K1A_BURST=$b timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q > $OUT/t_b$b.log 2>&1; echo "burst $b tests rc=$?"; tail -1 $OUT/t_b$b.log
done
for b in 1 21 22 2; do
  SBX_K1A_BURST=$b timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_b$b.json 2> $OUT/bench_b$b.err
  echo "burst $b rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench_b$b.json"))
print("K1a burst $b:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
PY
done
cd /tmp && export TMPDIR=/tmp
for b in 21 22 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    SBX_K1A_BURST=$b timeout 300 rocprofv3 --pmc $c --kernel-include-regex huffman --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p_${b}_$c -o p -- \
        python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/p_${b}_$c.err
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for b in (21, 22, 2):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("gpurun_out/r3l/p_%d_%s/**/*counter_collection.csv" % (b, c), recursive=True):
            best = 0.0
            for row in csv.DictReader(open(f)):
                if "huffman" in row.get("Kernel_Name", ""):
                    best = max(best, float(row["Counter_Value"]))
            print("K1a burst", b, c, "%.2f GB (raw)" % (best * 1024 / 1e9))
PY
find $OUT -name '*counter_collection.csv' -size +2M -delete
