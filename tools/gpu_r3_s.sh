#!/bin/bash
# round 3, a last short check: a BAM above the pipeline threshold through the CLI as one process (one-pass form) and detached (pipeline)
mkdir -p gpurun_out/r3s
tools/gen_bam --out /dev/shm/s.bam --contigs c1:15000000 --coverage 30 --seed 3 > /dev/null 2>&1; ls -la /dev/shm/s.bam | awk '{print $5}'
SBX_TIMING=1 SBX_NO_DETACH=1 sambamba_amd/csrc/sbx-depth base /dev/shm/s.bam 2> gpurun_out/r3s/single.err | md5sum
SBX_TIMING=1 sambamba_amd/csrc/sbx-depth base /dev/shm/s.bam 2> gpurun_out/r3s/detached.err | md5sum
grep "sbx-depth\] open" gpurun_out/r3s/single.err | cut -c1-160; grep "sbx-depth\] open" gpurun_out/r3s/detached.err | cut -c1-160
