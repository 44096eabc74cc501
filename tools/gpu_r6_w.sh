#!/bin/bash
# round 6, GPU call W (the final set of the round): the measurement set of config 2 on the round's kernels: counter passes stamped with the kernel sources (bench.py joins
# them into the line), the driver's invocation, smoke, then the whole GPU test-suite with all durations
OUT=$(pwd)/gpurun_out/r6_w
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 bash tools/pmc_pass.sh $OUT 2 2>&1 | tail -14
cp $OUT/pmc_fetch_write_config2.csv $OUT/pmc_sq_config2.csv profiles/round6/ 2>/dev/null
timeout 700 python bench.py --steps 20 --warmup 5 > $OUT/bench_config2_driver_invocation.json 2> $OUT/bench_config2_driver_invocation.err
echo "bench rc=$?"; tail -4 $OUT/bench_config2_driver_invocation.err
python - $OUT/bench_config2_driver_invocation.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], "Mreads/s", d["ms_per_step"], "ms", {k: v["ms"] for k, v in d["kernels"].items()})
r = d["roofline"]
print("roofline", {k: r.get(k) for k in ("kernel", "frac", "path_frac", "path_frac_incl_counters", "traffic", "traffic_over_algorithmic", "traffic_total", "path_traffic_over_algorithmic")}, r.get("issue_roofline"))
print("parity", d["parity_checked"]["ok"], d["parity_checked"].get("coverage"), "cpu", d["cpu_baseline"]["value"], "e2e", d["e2e"]["seconds"], d["e2e"]["all_seconds"], d["e2e"]["detached_seconds"])
print("device_text", d.get("device_text"))
PY
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 1100 python -m pytest tests -m gpu -q --durations=0 > $OUT/gpu_tests_full.log 2>&1
tail -3 $OUT/gpu_tests_full.log
grep -E "^[0-9.]+s (call|setup)" $OUT/gpu_tests_full.log | awk '{split($3,a,"::"); t[a[1]]+=$1} END {for (f in t) printf "%8.1f %s\n", t[f], f}' | sort -rn | head -30 | tee $OUT/gpu_tests_by_file.txt
