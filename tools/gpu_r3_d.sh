#!/bin/bash
# round 3, GPU call D: K3 variant 2, -m fast path, scans, describe occupancy, pipelined CLI: tests, then A/B timings
OUT=gpurun_out/r3d
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_depth.py tests/test_gpu_edge_cases.py tests/test_gpu_random_differential.py tests/test_gpu_mates.py tests/test_gpu_pipeline.py tests/test_gpu_batches.py tests/test_gpu_large_properties.py tests/test_gpu_region_window.py tests/test_gpu_multibam.py tests/test_gpu_repair.py -x -q > $OUT/t_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -c 1500 $OUT/t_kernels.log
SBX_K3_VARIANT=1 SBX_K7_VARIANT=0 timeout 600 python -m pytest tests/test_gpu_depth.py tests/test_gpu_mates.py -x -q > $OUT/t_old_variants.log 2>&1; echo "old variants rc=$?"; tail -2 $OUT/t_old_variants.log
export SBX_TIMING=1
for v in 2 1; do
  SBX_K3_VARIANT=$v timeout 600 python bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 6 > $OUT/bench_k3_v${v}.json 2> $OUT/bench_k3_v${v}.err
  echo "K3 variant $v rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench_k3_v${v}.json"))
print("K3 variant $v", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"], {k:v["frac_of_hbm_peak"] for k,v in d["kernels"].items()})
PY
done
BAM=$(ls /dev/shm/sbx_bench_*.bam | head -1)
for i in 1 2 3; do s=$(date +%s.%N); sambamba_amd/csrc/sbx-depth base -o /dev/null $BAM 2> $OUT/e2e_pipe_$i.err; e=$(date +%s.%N); echo "pipelined(4) $(echo "$e - $s" | bc) s" >> $OUT/e2e_runs.txt; grep "sbx-depth" $OUT/e2e_pipe_$i.err | tail -2 >> $OUT/e2e_runs.txt; sleep 2; done
for n in 2 8; do s=$(date +%s.%N); SBX_SLICE_POSITIONS=$((248956422 / n + 1)) sambamba_amd/csrc/sbx-depth base -o /dev/null $BAM 2> $OUT/e2e_pipe_n$n.err; e=$(date +%s.%N); echo "pipelined($n) $(echo "$e - $s" | bc) s" >> $OUT/e2e_runs.txt; grep "sbx-depth" $OUT/e2e_pipe_n$n.err | tail -2 >> $OUT/e2e_runs.txt; sleep 2; done
for i in 1 2; do s=$(date +%s.%N); SBX_NO_PIPELINE=1 sambamba_amd/csrc/sbx-depth base -o /dev/null $BAM 2> $OUT/e2e_one_$i.err; e=$(date +%s.%N); echo "one pass $(echo "$e - $s" | bc) s" >> $OUT/e2e_runs.txt; grep "sbx-depth" $OUT/e2e_one_$i.err | tail -2 >> $OUT/e2e_runs.txt; sleep 2; done
cat $OUT/e2e_runs.txt
# config 5 (-m -q20, 100 M reads): fast path on / off
for v in 1 0; do
  SBX_K7_VARIANT=$v timeout 900 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 6 > $OUT/bench_c5_k7v${v}.json 2> $OUT/bench_c5_k7v${v}.err
  echo "config 5 K7 variant $v rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench_c5_k7v${v}.json"))
print("c5 K7 variant $v", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"])
PY
done
