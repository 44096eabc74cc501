#!/bin/bash
# round 3, GPU call G: K1a variants (burst depth / tuned decode), upload pool, detached teardown
OUT=gpurun_out/r3g
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_inflate.py -x -q > $OUT/t_default.log 2>&1; echo "default tests rc=$?"; tail -2 $OUT/t_default.log
for v in 5 6 2; do
  SBX_K1A_VARIANT=$v timeout 400 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_edge_cases.py tests/test_gpu_depth.py -x -q > $OUT/t_k1a$v.log 2>&1; echo "K1a variant $v tests rc=$?"; tail -1 $OUT/t_k1a$v.log
done
export SBX_TIMING=1
for v in 1 2 3 4 5 6; do
  SBX_K1A_VARIANT=$v timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 6 > $OUT/bench_k1a$v.json 2> $OUT/bench_k1a$v.err
  echo "K1a $v rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_k1a$v.json"))
    print("K1a variant $v:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as e: print("K1a variant $v: no line", e)
PY
done
BAM=$(ls /dev/shm/sbx_bench_*.bam | head -1)
# HBM write traffic of the token stores per burst depth (counters in their own runs, K1a only)
cd /tmp && export TMPDIR=/tmp
for v in 1 5 6; do
  SBX_K1A_VARIANT=$v timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex huffman --output-format csv -d $GRAFT_REPO_ROOT/$OUT/w$v -o w -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/w$v.err
  echo "pmc write K1a $v rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for v in (1, 5, 6):
    for f in glob.glob("gpurun_out/r3g/w%d/**/*counter_collection.csv" % v, recursive=True):
        tot = {}
        for row in csv.DictReader(open(f)):
            if "huffman" in row.get("Kernel_Name", ""):
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        print("K1a variant", v, "raw counters per run:", tot)
PY
find $OUT -name '*counter_collection.csv' -size +4M -delete
now() { python -c 'import time; print(time.time())'; }
for tag in "detached:" "single:SBX_NO_DETACH=1" "thr4:SBX_UPLOAD_THREADS=4" "thr8:SBX_UPLOAD_THREADS=8" "thr16:SBX_UPLOAD_THREADS=16" "onepass:SBX_NO_PIPELINE=1"; do
  name=${tag%%:*}; envs=${tag#*:}
  for i in 1 2 3; do s=$(now); env $envs sambamba_amd/csrc/sbx-depth base -o /dev/null $BAM 2> $OUT/e2e_${name}_$i.err; e=$(now); python -c "print('$name wall %.3f s' % ($e - $s))" >> $OUT/e2e_runs.txt; grep "sbx-depth\] open\|batch refs" $OUT/e2e_${name}_$i.err | tail -2 | cut -c1-230 >> $OUT/e2e_runs.txt; sleep 2; done
done
cat $OUT/e2e_runs.txt
# fixed cost of a CLI run on a tiny input, both process models
for tag in "tiny_detached:" "tiny_single:SBX_NO_DETACH=1"; do
  name=${tag%%:*}; envs=${tag#*:}
  for i in 1 2 3; do s=$(now); env $envs sambamba_amd/csrc/sbx-depth base -o /dev/null tests/golden/issue_193.bam 2>/dev/null; e=$(now); python -c "print('$name wall %.3f s' % ($e - $s))" >> $OUT/e2e_runs.txt; sleep 1; done
done
tail -6 $OUT/e2e_runs.txt
