#!/bin/bash
# round 3, GPU call E: K1b variants (tests + timing), pipelined e2e after the text-buffer fix, generator through the device writer
OUT=gpurun_out/r3e
mkdir -p $OUT
for v in 1 2 3; do
  SBX_K1B_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_depth.py -x -q > $OUT/t_k1b_v$v.log 2>&1; echo "K1b variant $v tests rc=$?"; tail -2 $OUT/t_k1b_v$v.log
done
timeout 900 python -m pytest tests/test_gpu_writer.py -x -q -k "generator or many_blocks or round_trip" > $OUT/t_writer.log 2>&1; echo "writer rc=$?"; tail -3 $OUT/t_writer.log
export SBX_TIMING=1
for v in 0 1 2 3; do
  SBX_K1B_VARIANT=$v timeout 600 python bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-e2e --parity-windows 6 > $OUT/bench_k1b_v${v}.json 2> $OUT/bench_k1b_v${v}.err
  echo "K1b variant $v rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench_k1b_v${v}.json"))
print("K1b variant $v", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"])
PY
done
BAM=$(ls /dev/shm/sbx_bench_*.bam | head -1)
now() { python -c 'import time; print(time.time())'; }
for tag in "pipelined:" "pipelined_piece8M:SBX_STREAM_PIECE=8388608" "pipelined_piece1M:SBX_STREAM_PIECE=1048576" "onepass:SBX_NO_PIPELINE=1"; do
  name=${tag%%:*}; envs=${tag#*:}
  for i in 1 2; do s=$(now); env $envs sambamba_amd/csrc/sbx-depth base -o /dev/null $BAM 2> $OUT/e2e_${name}_$i.err; e=$(now); python -c "print('$name wall %.3f s' % ($e - $s))" >> $OUT/e2e_runs.txt; grep "sbx-depth" $OUT/e2e_${name}_$i.err | tail -2 >> $OUT/e2e_runs.txt; sleep 2; done
done
cat $OUT/e2e_runs.txt
# the generator through the device writer: chr1 at 30x (time, size), then the headline pipeline on that file
s=$(now); GEN_TIMING=1 tools/gen_bam --out /dev/shm/dev_chr1.bam --contigs chr1:248956422 --coverage 30 --seed 0x5A4D0002 --codec device > $OUT/gen_device.json 2> $OUT/gen_device.err; e=$(now)
python -c "print('gen_bam --codec device chr1: %.1f s' % ($e - $s))"; cat $OUT/gen_device.json; tail -2 $OUT/gen_device.err; ls -la /dev/shm/dev_chr1.bam
python - <<PY
import sambamba_amd, json
d=sambamba_amd.Depth("/dev/shm/dev_chr1.bam"); d.set_params(); d.preload()
for i in range(3): st=d.run()
print("device-written chr1:", {k: round(st[k],2) for k in ("ms_huffman","ms_lz77","ms_index","ms_accumulate","ms_total")}, st["n_records"], st["compressed_bytes"], st["token_bytes"])
PY
du -sh gpurun_out/* | tail -5
