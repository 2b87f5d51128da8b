#!/bin/bash
# LU timing with different numbers of CUs set aside for the panel stream.
set -o pipefail
mkdir -p gpurun_out
{
for c in ${@:-16 24 32 40}; do
  echo "== EIGAMD_LU_PANEL_CUS=$c"
  EIGAMD_LU_PANEL_CUS=$c timeout -k 10 120 python -u tools/lu_time.py 16384 32768 || exit 1
done
} 2>&1 | tee gpurun_out/lu_cus.log
