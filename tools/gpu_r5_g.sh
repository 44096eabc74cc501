#!/bin/bash
# round 5, GPU call G: every window of a run in two launches (window cache) -- tests; config 3 at 2 % with and without it;
# config 2 end to end as one process by pipeline form and slice count
set -u
OUT=gpurun_out/r5_g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_windows.py tests/test_gpu_region_window.py tests/test_gpu_compact.py tests/test_gpu_batches.py tests/test_gpu_multibam.py tests/test_gpu_mates.py -q -x 2>&1 | tail -5 | tee $OUT/tests_windows.txt
timeout 600 python -m pytest tests/test_gpu_dist.py -q -x -k "sharded_equals or single_contig or config" 2>&1 | tail -4 | tee $OUT/tests_dist.txt
for m in 1 0; do
  SBX_WINDOWS_AT_ONCE=$m timeout 300 python bench.py --config 3 --scale 0.02 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 8 > $OUT/bench_config3_scale002_atonce$m.json 2> /tmp/c3_$m.err
  python - $OUT/bench_config3_scale002_atonce$m.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], d["value"], d["ms_per_step"], "kernels sum", round(sum(v["ms"] for v in d["kernels"].values()), 2), d["parity_checked"]["ok"], (d["parity_checked"].get("whole_contig") or {}).get("ok"))
except Exception as e:
    print("no line", e)
PY
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-full-parity --parity-windows 0 > /dev/null 2> /tmp/gen.err    # (generates the chr1 BAM)
BAM=$(ls -S /dev/shm/sbx_bench_*.bam | head -1)
ls -la $BAM
CLI=sambamba_amd/csrc/sbx-depth
run() {   # label, env assignments...
  local label=$1; shift
  for i in 1 2 3; do
    sleep 2
    local t0=$(date +%s.%N)
    env SBX_TIMING=1 "$@" $CLI base -o /dev/null $BAM 2> /tmp/cli.err
    local t1=$(date +%s.%N)
    echo "$label wall $(python -c "print(round($t1-$t0,3))") s | $(grep -E 'slices through|total' /tmp/cli.err | tail -1 | cut -c1-260)"
  done
}
{
run "one-pass" SBX_NO_PIPELINE=1
run "1ctx-4slices" SBX_X=1
run "1ctx-8slices" SBX_SLICE_POSITIONS=31200000
run "1ctx-16slices" SBX_SLICE_POSITIONS=15600000
run "2ctx-4slices" SBX_PIPELINE_CONTEXTS=2
run "detached" SBX_DETACH=1
} | tee $OUT/e2e_config2.txt
