#!/bin/bash
# round 3, GPU call A: box facts, config 2 headline with the fast generator, configs 3 and 4 at full scale
OUT=gpurun_out/r3a
mkdir -p $OUT
{
  echo "nproc $(nproc)"; free -g | head -2; df -h /dev/shm /tmp | cat; lscpu | grep -E "Model name|Socket|Thread|Core" ; 
  rocm-smi --showmeminfo vram 2>/dev/null | grep -i total | head -2
} > $OUT/box.txt 2>&1
cat $OUT/box.txt
export SBX_TIMING=1
timeout 600 python bench.py --steps 20 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
echo "c2 rc=$?"; tail -c 600 $OUT/bench_c2.err
avail=$(awk '/MemAvailable/ {print int($2/1048576)}' /proc/meminfo)
shm=$(df -BG --output=avail /dev/shm | tail -1 | tr -dc 0-9)
echo "avail ${avail} GB, shm ${shm} GB"
if [ "$avail" -gt 260 ] && [ "$shm" -gt 100 ]; then
  timeout 1200 python bench.py --config 3 --steps 2 --warmup 1 --parity-windows 18 --cpu-sample-reads 1000000 > $OUT/bench_c3_full.json 2> $OUT/bench_c3_full.err
  echo "c3 rc=$?"; tail -c 1500 $OUT/bench_c3_full.err
  timeout 900 python bench.py --config 4 --steps 3 --warmup 1 --parity-windows 16 --cpu-sample-reads 1000000 > $OUT/bench_c4_full.json 2> $OUT/bench_c4_full.err
  echo "c4 rc=$?"; tail -c 1500 $OUT/bench_c4_full.err
else
  echo "not enough host memory for the whole-genome BAM: skipped" | tee $OUT/skipped.txt
fi
ls -la /dev/shm | head; 
