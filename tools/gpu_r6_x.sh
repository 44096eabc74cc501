#!/bin/bash
# round 6, GPU call X: BASELINE configs 5, 3 and 4 at FULL scale on the round's final code (the lines; their counter passes are call M's, of
# the code before the K2 / K6 work)
OUT=$(pwd)/gpurun_out/r6_x
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in 5 3 4; do
  timeout 1700 python bench.py --config $cfg --steps 5 --warmup 1 --parity-windows 32 > $OUT/bench_config${cfg}_full.json 2> /tmp/bench_config$cfg.err
  echo "config $cfg rc=$?"; tail -c 300 /tmp/bench_config$cfg.err | tr '\n' ' '; echo
  python - $OUT/bench_config${cfg}_full.json $cfg <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    pc = d["parity_checked"]
    print("config", sys.argv[2], ":", d["value"], "Mreads/s", d["ms_per_step"], "ms", {k: round(v["ms"], 2) for k, v in d["kernels"].items()},
          "parity", pc.get("ok"), pc.get("windows"), "whole", {k: v for k, v in (pc.get("whole_contig") or pc.get("full_text") or pc).items() if k in ("coverage", "regions", "ok", "contig", "text_bytes")},
          "cpu", (d.get("cpu_baseline") or {}).get("value"), "e2e", (d.get("e2e") or {}).get("seconds"), (d.get("e2e") or {}).get("detached_seconds"),
          "rerun_cached", (d.get("rerun_cached") or {}).get("ms_per_step"), "device_text", (d.get("device_text") or {}).get("ms_per_step"))
except Exception as e:
    print("no line", e)
PY
done
