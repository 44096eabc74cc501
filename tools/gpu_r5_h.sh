#!/bin/bash
# round 5, GPU call H: BASELINE configs 3, 4 (whole genome, zlib-6 input generated once for both) and 5 at FULL scale as measured lines:
# cpu_baseline, e2e, sampled + whole-contig parity, 5 steps; then the stamped counter passes of each (tools/pmc_pass.sh).
OUT=$(pwd)/gpurun_out/r5_h
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in 3 4 5; do
  timeout 1500 python bench.py --config $cfg --steps 5 --warmup 1 --parity-windows 32 > $OUT/bench_config${cfg}_full.json 2> /tmp/bench_config$cfg.err
  echo "config $cfg rc=$?"; tail -c 300 /tmp/bench_config$cfg.err | tr '\n' ' '; echo
  python - $OUT/bench_config${cfg}_full.json $cfg <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    pc = d["parity_checked"]
    print("config", sys.argv[2], ":", d["value"], "Mreads/s", d["ms_per_step"], "ms", {k: v["ms"] for k, v in d["kernels"].items()}, "sum", round(sum(v["ms"] for v in d["kernels"].values()), 1),
          "parity", pc.get("ok"), pc.get("windows"), "whole", (pc.get("whole_contig") or pc.get("full_text") or {}), "cpu", (d.get("cpu_baseline") or {}).get("value"),
          "e2e", (d.get("e2e") or {}).get("seconds"), (d.get("e2e") or {}).get("detached_seconds"), d["host"])
except Exception as e:
    print("no line", e)
PY
done
for cfg in 3 4 5; do
  timeout 600 bash tools/pmc_pass.sh $OUT $cfg 2>&1 | tail -8
done
du -sh $OUT
