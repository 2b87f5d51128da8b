#!/bin/bash
# LU timing with the panel stream updating the last N columns of the trailing matrix itself (0: off)
# while at least M columns remain.
set -o pipefail
mkdir -p gpurun_out
{
for c in ${@:-"0 0" "768 20480" "1024 20480" "1024 24576" "1280 22528" "768 18432"}; do
  set -- $c
  echo "== EIGAMD_LU_SLICE=$1 EIGAMD_LU_SLICE_MIN=$2"
  EIGAMD_LU_SLICE=$1 EIGAMD_LU_SLICE_MIN=$2 timeout -k 10 120 python -u tools/lu_time.py 24576 32768 || exit 1
done
} 2>&1 | tee gpurun_out/lu_slice.log
