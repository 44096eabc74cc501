#!/bin/bash
# round 4, call R: the level-aware BGZF writer on the device (dynamic Huffman codes) -- parity with the host build of the same
# encoder at every level, then the kernel's rate on 16384 blocks (1.07 GB of BAM-like stream) per mode.
set -u
mkdir -p gpurun_out
timeout 110 python -m pytest tests/test_gpu_writer.py -x -q -k "host_encoder or default_level or trade" 2>&1 | tail -6 | tee gpurun_out/r_writer_tests.txt
SBX_TIMING=1 timeout 80 python - 2>&1 <<'PY' | grep -E "bgzf_compress|level|Error|error" | tee gpurun_out/r_writer_rate.txt
import sys, time
import sambamba_amd
from tests.test_deflate_core_cpu import bam_like
unit = bam_like(1_000_000, 5)
data = unit * (16384 * 0xFF00 // len(unit) + 1)
data = data[:16384 * 0xFF00]
for lv in (1, 6, 9):
    t = time.time()
    c = sambamba_amd.bgzf_compress(data, level=lv)
    print("level %d: %d -> %d bytes (ratio %.3f) in %.2f s wall" % (lv, len(data), len(c), len(data) / len(c), time.time() - t), flush=True)
PY
