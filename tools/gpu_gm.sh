#!/bin/bash
# Timing of f64 products of several shapes for two group sizes of the tile map.
set -o pipefail
mkdir -p gpurun_out
{
for gm in 0 8; do
  echo "== EIGAMD_GROUP_M=$gm (0: the kind's default)"
  for shape in "16384 16384 16384" "32768 32768 256" "16384 16384 256" "32768 32768 1024" "8192 8192 8192" "4096 4096 4096"; do
    EIGAMD_GROUP_M=$gm timeout -k 10 120 ./build/gemm_check f64 N N $shape 5 0 -1 0 1 0 | grep -o "m=.*best=.*" || exit 1
  done
  EIGAMD_GROUP_M=$gm timeout -k 10 120 ./build/gemm_check f64 T N 16384 16384 16384 3 0 1 0 1 0 | grep -o "m=.*best=.*" || exit 1
  EIGAMD_GROUP_M=$gm timeout -k 10 200 python -u tools/lu_time.py 16384 32768 || exit 1
done
} 2>&1 | tee gpurun_out/gm.log
