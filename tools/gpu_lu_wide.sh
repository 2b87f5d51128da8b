#!/bin/bash
# LU parity tests, then LU timing for several settings of the outer step width:
# arguments are "T4:T2" threshold pairs (EIGAMD_LU_WIDE_T4 / _T2) or "wN" for a fixed width N.
set -o pipefail
mkdir -p gpurun_out
{
timeout -k 10 300 python -u -m pytest tests/test_lu_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread || exit 1
for a in "$@"; do
  echo "== $a"
  if [[ $a == w* ]]; then
    EIGAMD_LU_WIDE=${a#w} timeout -k 10 150 python -u tools/lu_time.py 16384 32768 || exit 1
  else
    EIGAMD_LU_WIDE_T4=${a%%:*} EIGAMD_LU_WIDE_T2=${a##*:} timeout -k 10 150 python -u tools/lu_time.py 16384 32768 || exit 1
  fi
done
} 2>&1 | tee gpurun_out/lu_wide.log
