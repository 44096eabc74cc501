#!/bin/bash
# round 3, GPU call C: writer + pipeline tests, PCIe rates by NUMA node, config 2 (K2 stats fix, pipelined e2e), config 4 host overhead
OUT=gpurun_out/r3c
mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 -o /tmp/pcie_bw tools/pcie_bw.cpp 2> $OUT/pcie_build.err && timeout 120 /tmp/pcie_bw > $OUT/pcie_bw.txt 2>&1; cat $OUT/pcie_bw.txt
timeout 900 python -m pytest tests/test_gpu_writer.py tests/test_gpu_pipeline.py -x -q > $OUT/t_writer_pipeline.log 2>&1; echo "writer+pipeline rc=$?"; tail -c 2500 $OUT/t_writer_pipeline.log
timeout 900 python -m pytest tests/test_gpu_region_window.py tests/test_gpu_windows.py tests/test_gpu_depth.py tests/test_gpu_worklist.py -x -q > $OUT/t_misc.log 2>&1; echo "misc rc=$?"; tail -3 $OUT/t_misc.log
export SBX_TIMING=1
timeout 600 python bench.py --steps 20 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"; tail -c 900 $OUT/bench_c2.err
python - <<PY
import json
d=json.load(open("$OUT/bench_c2.json"))
print(d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["parity_checked"]["ok"], "e2e", d["e2e"])
PY
# the same BAM through the CLI with and without the pipeline, a few runs each
BAM=$(ls /dev/shm/sbx_bench_*.bam | head -1)
for i in 1 2 3; do /usr/bin/time -f "pipelined %e s" sambamba_amd/csrc/sbx-depth base -o /dev/null $BAM 2>> $OUT/e2e_runs.txt; sleep 2; done
for i in 1 2; do SBX_NO_PIPELINE=1 /usr/bin/time -f "one pass %e s" sambamba_amd/csrc/sbx-depth base -o /dev/null $BAM 2>> $OUT/e2e_runs.txt; sleep 2; done
for n in 4 16; do SBX_SLICE_POSITIONS=$((248956422 / n + 1)) /usr/bin/time -f "pipelined $n slices %e s" sambamba_amd/csrc/sbx-depth base -o /dev/null $BAM 2>> $OUT/e2e_runs.txt; sleep 2; done
grep -v "^\[sbx\] hipMalloc" $OUT/e2e_runs.txt | tail -30
sambamba_amd/csrc/sbx-depth base $BAM | md5sum > $OUT/md5_pipelined.txt; SBX_NO_PIPELINE=1 sambamba_amd/csrc/sbx-depth base $BAM | md5sum > $OUT/md5_onepass.txt; cat $OUT/md5_*.txt
