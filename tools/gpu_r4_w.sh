#!/bin/bash
# round 4, call W: which consumer builds the index (the device by default), and what it is worth on a 20-Mbp BAM
set -u
mkdir -p gpurun_out
timeout 60 python -m pytest tests/test_gpu_writer.py -x -q -k "consumed_on_the_device" 2>&1 | tail -5 | tee gpurun_out/w_consumer_test.txt
tools/gen_bam --out /tmp/w.bam --contigs chrW:20000000 --coverage 30 --codec device > /dev/null 2>&1
for mode in device host; do
  if [ $mode = host ]; then export SBX_BAI_HOST=1; fi
  SBX_TIMING=1 python - 2>&1 <<'PY' | grep -E "build_index|wall" | tee -a gpurun_out/w_index_rate.txt
import time, os, sambamba_amd
t = time.time(); sambamba_amd.build_index("/tmp/w.bam", "/tmp/w.%s.bai" % ("host" if os.environ.get("SBX_BAI_HOST") else "dev"))
print("wall %.3f s (%s)" % (time.time() - t, "SBX_BAI_HOST=1" if os.environ.get("SBX_BAI_HOST") else "default"))
PY
done
cmp /tmp/w.dev.bai /tmp/w.host.bai && echo "same index" | tee -a gpurun_out/w_index_rate.txt
