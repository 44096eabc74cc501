#!/bin/bash
# round 5, the final set on the code of the round's last kernel change: counter passes of config 2 stamped with the kernel sources
# (bench.py joins them into the line), the driver's invocation, smoke, then the whole GPU test-suite.
OUT=$(pwd)/gpurun_out/r5_final
mkdir -p $OUT
export TMPDIR=/tmp
timeout 700 bash tools/pmc_pass.sh $OUT 2 2>&1 | tail -14
cp $OUT/pmc_fetch_write_config2.csv $OUT/pmc_sq_config2.csv profiles/round5/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_config2_driver_invocation.json 2> $OUT/bench_config2_driver_invocation.err
echo "bench rc=$?"; tail -4 $OUT/bench_config2_driver_invocation.err
python - $OUT/bench_config2_driver_invocation.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], "Mreads/s", d["ms_per_step"], "ms", {k: v["ms"] for k, v in d["kernels"].items()})
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "path_frac", "traffic", "traffic_over_algorithmic")}, d["roofline"].get("issue_roofline"))
print("parity", d["parity_checked"]["ok"], d["parity_checked"].get("coverage"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["runs"], "e2e", d["e2e"]["seconds"], d["e2e"]["all_seconds"], d["e2e"]["detached_seconds"])
PY
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 840 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -22 | tee $OUT/gpu_tests.log
