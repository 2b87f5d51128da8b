#!/bin/bash
# LU timing with and without the narrow first outer steps.
set -o pipefail
mkdir -p gpurun_out
{
for c in ${@:-0 1}; do
  echo "== EIGAMD_LU_RAMP=$c"
  EIGAMD_LU_RAMP=$c timeout -k 10 120 python -u tools/lu_time.py 16384 24576 32768 || exit 1
done
} 2>&1 | tee gpurun_out/lu_ramp.log
