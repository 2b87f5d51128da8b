#!/bin/bash
# LU timing with the bulk update on one stream (0) or dealt over two while at least N columns remain.
set -o pipefail
mkdir -p gpurun_out
{
for c in ${@:-0 8192 4096 16384}; do
  echo "== EIGAMD_LU_SPLIT_MIN=$c"
  EIGAMD_LU_SPLIT_MIN=$c timeout -k 10 120 python -u tools/lu_time.py 16384 32768 || exit 1
done
} 2>&1 | tee gpurun_out/lu_split.log
