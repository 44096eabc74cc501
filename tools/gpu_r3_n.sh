#!/bin/bash
# round 3, GPU call N: the N > 1 line of bench.py at full size, two ranks sharing the box's one GPU (gloo): not a scaling number --
# the sharded line, the parity of both ranks, the strong (one contig cut in two) and all-reduce side measurements
OUT=gpurun_out/r3n
mkdir -p $OUT
SBX_BENCH_BACKEND=gloo timeout 560 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err; echo "2 ranks rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r3n/bench_2ranks.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["scaling"], d["config"]["sharding"][:80], d["parity_checked"].get("ok"), "strong:", json.dumps(d.get("strong_one_contig"))[:400], "allreduce:", json.dumps(d.get("allreduce_option"))[:400])
except Exception as e: print("no line", e)
PY
tail -c 1200 $OUT/bench_2ranks.err
