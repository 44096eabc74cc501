#!/bin/bash
# End-to-end wall time of the product CLI (file -> text) with device-formatted rows and with the host
# emulation of PerBasePrinter, plus the CPU oracle on a sample, on a synthetic BAM of the given length.
set -u
LEN=${1:-90000000}
BAM=/dev/shm/sbx_e2e_$LEN.bam
[ -f $BAM ] || tools/gen_bam --out $BAM --contigs chr1:$LEN --coverage 30 --seed 1515847681 --threads $(nproc) > /dev/null
CLI=sambamba_amd/csrc/sbx-depth
for mode in device host; do
  if [ $mode = host ]; then export SBX_HOST_FORMAT=1; else unset SBX_HOST_FORMAT; fi
  s=$(date +%s%N)
  $CLI base $BAM | wc -c > /tmp/e2e_bytes
  e=$(date +%s%N)
  echo "cli_e2e mode=$mode length=$LEN seconds=$(( (e - s) / 1000000 ))e-3 bytes=$(cat /tmp/e2e_bytes)"
done
unset SBX_HOST_FORMAT
$CLI base $BAM | md5sum
SBX_HOST_FORMAT=1 $CLI base $BAM | md5sum
s=$(date +%s%N); $CLI base $BAM > /dev/null; e=$(date +%s%N); echo "cli_e2e mode=device to=/dev/null seconds=$(( (e - s) / 1000000 ))e-3"
s=$(date +%s%N); ORC_STATS=1 oracle/depth_oracle base --max-reads 2000000 $BAM > /dev/null; e=$(date +%s%N); echo "oracle 2M reads seconds=$(( (e - s) / 1000000 ))e-3"
