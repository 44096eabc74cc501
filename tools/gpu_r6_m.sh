#!/bin/bash
# round 6, GPU call M: the counter passes of config 2 again (ONE pass per counter run: the side measurements off), then BASELINE configs 3, 4
# (whole genome, zlib-6 input generated once for both) and 5 at FULL scale as measured lines with their stamped counter passes
OUT=$(pwd)/gpurun_out/r6_m
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 bash tools/pmc_pass.sh $OUT 2 2>&1 | tail -9
for cfg in 5 3 4; do
  timeout 1700 python bench.py --config $cfg --steps 5 --warmup 1 --parity-windows 32 > $OUT/bench_config${cfg}_full.json 2> /tmp/bench_config$cfg.err
  echo "config $cfg rc=$?"; tail -c 300 /tmp/bench_config$cfg.err | tr '\n' ' '; echo
  python - $OUT/bench_config${cfg}_full.json $cfg <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    pc = d["parity_checked"]
    print("config", sys.argv[2], ":", d["value"], "Mreads/s", d["ms_per_step"], "ms", {k: v["ms"] for k, v in d["kernels"].items()}, "sum", round(sum(v["ms"] for v in d["kernels"].values()), 1),
          "parity", pc.get("ok"), pc.get("windows"), "whole", (pc.get("whole_contig") or pc.get("full_text") or {}), "cpu", (d.get("cpu_baseline") or {}).get("value"),
          "e2e", (d.get("e2e") or {}).get("seconds"), (d.get("e2e") or {}).get("detached_seconds"), "rerun_cached", d.get("rerun_cached"), "device_text", d.get("device_text"))
except Exception as e:
    print("no line", e)
PY
  timeout 700 bash tools/pmc_pass.sh $OUT $cfg 2>&1 | tail -6
done
du -sh $OUT
