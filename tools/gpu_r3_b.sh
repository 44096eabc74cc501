#!/bin/bash
# round 3, GPU call B: new multi-rank / writer / index tests, K1a variant A/B on the headline workload
OUT=gpurun_out/r3b
mkdir -p $OUT
{ cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -6; grep -i cpus_allowed_list /proc/self/status; } > $OUT/cgroup.txt 2>&1
cat $OUT/cgroup.txt
timeout 900 python -m pytest tests/test_gpu_writer.py tests/test_gpu_inflate.py -x -q > $OUT/t_writer_inflate.log 2>&1; echo "writer+inflate rc=$?"; tail -5 $OUT/t_writer_inflate.log
SBX_K1A_VARIANT=0 timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q > $OUT/t_inflate_v0.log 2>&1; echo "inflate v0 rc=$?"; tail -2 $OUT/t_inflate_v0.log
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q -k "writes_its_own or allreduce or rccl or owned or beyond or mate_slack or bench_multi or base_on_a_genome" > $OUT/t_dist_new.log 2>&1; echo "dist rc=$?"; tail -12 $OUT/t_dist_new.log
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q > $OUT/t_bench.log 2>&1; echo "bench tests rc=$?"; tail -5 $OUT/t_bench.log
for v in 1 0 1; do
  SBX_K1A_VARIANT=$v timeout 600 python bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 4 > $OUT/bench_k1a_v${v}.json 2> $OUT/bench_k1a_v${v}.err
  echo "variant $v rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench_k1a_v${v}.json"))
print("variant $v", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
PY
done
