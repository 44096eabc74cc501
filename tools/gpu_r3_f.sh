#!/bin/bash
# round 3, GPU call F: K1b variant with 8 waves, K3 variant 3, exit cost of the CLI
OUT=gpurun_out/r3f
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_depth.py tests/test_gpu_edge_cases.py tests/test_gpu_random_differential.py tests/test_gpu_region_window.py -x -q > $OUT/t_default.log 2>&1; echo "default variants tests rc=$?"; tail -2 $OUT/t_default.log
export SBX_TIMING=1
for combo in "1 3" "0 3" "1 2"; do
  set -- $combo
  SBX_K1B_VARIANT=$1 SBX_K3_VARIANT=$2 timeout 600 python bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 6 > $OUT/bench_k1b$1_k3$2.json 2> $OUT/bench_k1b$1_k3$2.err
  echo "K1b $1 K3 $2 rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench_k1b$1_k3$2.json"))
print("K1b variant $1 K3 variant $2:", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
PY
done
BAM=$(ls /dev/shm/sbx_bench_*.bam | head -1)
now() { python -c 'import time; print(time.time())'; }
for i in 1 2 3; do s=$(now); sambamba_amd/csrc/sbx-depth > /dev/null 2>&1; e=$(now); python -c "print('usage only wall %.3f s' % ($e - $s))" >> $OUT/e2e_runs.txt; done
for tag in "pipelined:" "pipelined_orderly:SBX_ORDERLY_EXIT=1" "pipelined8:SBX_SLICE_POSITIONS=31119553" "pipelined8_orderly:SBX_SLICE_POSITIONS=31119553 SBX_ORDERLY_EXIT=1" "onepass_orderly:SBX_NO_PIPELINE=1 SBX_ORDERLY_EXIT=1"; do
  name=${tag%%:*}; envs=${tag#*:}
  for i in 1 2; do s=$(now); env $envs SBX_STREAM_PIECE=1048576 sambamba_amd/csrc/sbx-depth base -o /dev/null $BAM 2> $OUT/e2e_${name}_$i.err; e=$(now); python -c "print('$name wall %.3f s' % ($e - $s))" >> $OUT/e2e_runs.txt; grep "sbx-depth\] open" $OUT/e2e_${name}_$i.err | tail -1 | cut -c1-220 >> $OUT/e2e_runs.txt; sleep 2; done
done
cat $OUT/e2e_runs.txt
