#!/bin/bash
# round 5, GPU call F: the one-context slice pipeline of the CLI -- tests, then config 2 end to end (one process) by slice count
set -u
OUT=gpurun_out/r5_f
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q 2>&1 | tail -5 | tee $OUT/tests_pipeline.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-full-parity --parity-windows 0 > /dev/null 2> /tmp/gen.err    # (generates the BAM)
BAM=$(ls /dev/shm/sbx_bench_*.bam | head -1)
CLI=sambamba_amd/csrc/sbx-depth
run() {   # label, env...
  local label=$1; shift
  for i in 1 2 3; do
    sleep 2
    /usr/bin/time -f "$label wall %e s" env SBX_TIMING=1 "$@" $CLI base -o /dev/null $BAM 2>&1 | grep -E "wall|slices through|total" | tr '\n' ' '; echo
  done
}
run "one-pass" SBX_NO_PIPELINE=1 | tee $OUT/e2e_config2.txt
run "1ctx-4slices" SBX_X=1 | tee -a $OUT/e2e_config2.txt
run "1ctx-8slices" SBX_SLICE_POSITIONS=31200000 | tee -a $OUT/e2e_config2.txt
run "1ctx-16slices" SBX_SLICE_POSITIONS=15600000 | tee -a $OUT/e2e_config2.txt
run "2ctx-4slices" SBX_PIPELINE_CONTEXTS=2 | tee -a $OUT/e2e_config2.txt
run "detached" SBX_DETACH=1 | tee -a $OUT/e2e_config2.txt
