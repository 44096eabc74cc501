#!/bin/bash
# round 3, GPU call H: K1a with the packed symbol table / scalar flags, K1b with scalar block state -- tests and timings
OUT=gpurun_out/r3h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_edge_cases.py tests/test_gpu_depth.py tests/test_gpu_repair.py tests/test_gpu_writer.py tests/test_gpu_mates.py tests/test_gpu_random_differential.py -x -q > $OUT/t_default.log 2>&1; echo "default tests rc=$?"; tail -2 $OUT/t_default.log
for b in 2 4; do
  SBX_K1A_BURST=$b SBX_K1B_VARIANT=0 timeout 400 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_edge_cases.py -x -q > $OUT/t_burst$b.log 2>&1; echo "burst $b + old K1b tests rc=$?"; tail -1 $OUT/t_burst$b.log
done
export SBX_TIMING=1
for combo in "1 1" "2 1" "4 1" "1 0" "1 3"; do
  set -- $combo
  SBX_K1A_BURST=$1 SBX_K1B_VARIANT=$2 timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 6 > $OUT/bench_burst$1_k1b$2.json 2> $OUT/bench_burst$1_k1b$2.err
  echo "burst $1 K1b $2 rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_burst$1_k1b$2.json"))
    print("K1a burst $1, K1b variant $2:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
except Exception as e: print("no line", e)
PY
done
BAM=$(ls /dev/shm/sbx_bench_*.bam | head -1)
now() { python -c 'import time; print(time.time())'; }
for tag in "detached:" "single:SBX_NO_DETACH=1"; do
  name=${tag%%:*}; envs=${tag#*:}
  for i in 1 2 3; do s=$(now); env $envs sambamba_amd/csrc/sbx-depth base -o /dev/null $BAM 2> $OUT/e2e_${name}_$i.err; e=$(now); python -c "print('$name wall %.3f s' % ($e - $s))" >> $OUT/e2e_runs.txt; grep "sbx-depth\] open\|batch refs" $OUT/e2e_${name}_$i.err | tail -2 | cut -c1-230 >> $OUT/e2e_runs.txt; sleep 2; done
done
cat $OUT/e2e_runs.txt
