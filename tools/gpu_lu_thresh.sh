#!/bin/bash
# LU timing over the thresholds of the outer-step width rule.
set -o pipefail
mkdir -p gpurun_out
{
for t in "20480 16640" "16384 12288" "24576 16640" "20480 12288" "18432 14336" "24576 20480"; do
  set -- $t
  echo "== T4=$1 T2=$2"
  EIGAMD_LU_WIDE_T4=$1 EIGAMD_LU_WIDE_T2=$2 timeout -k 10 120 python -u tools/lu_time.py 24576 32768 || exit 1
done
} 2>&1 | tee gpurun_out/lu_thresh.log
