#!/bin/bash
# LU timing with different numbers of CUs set aside for the panel stream.
set -o pipefail
mkdir -p gpurun_out
{
echo "== masks on, small size first"
timeout -k 5 60 python -u tools/lu_time.py 2048 4096 || exit 1
timeout -k 10 200 python -u -m pytest tests/test_lu_gpu.py -x -q --timeout 120 --timeout-method thread || exit 1
for c in 0 32 64; do
  echo "== EIGAMD_LU_PANEL_CUS=$c"
  EIGAMD_LU_PANEL_CUS=$c timeout -k 10 120 python -u tools/lu_time.py 8192 16384 32768 || exit 1
done
} 2>&1 | tee gpurun_out/lu_cus.log
