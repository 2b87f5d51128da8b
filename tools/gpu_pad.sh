#!/bin/bash
# Effect of the leading dimension (power of two or not) and of beta on the short-reduction GEMM.
set -o pipefail
mkdir -p gpurun_out
{
for pad in 0 16 32 144; do
  timeout -k 10 120 ./build/gemm_check f64 N N 32768 32768 256 5 $pad -1 0 1 0 | grep -o "m=.*best=.*" || exit 1
done
echo "beta = 0 (no C read)"
timeout -k 10 120 ./build/gemm_check f64 N N 32768 32768 256 5 0 -1 0 0 0 | grep -o "m=.*best=.*" || exit 1
timeout -k 10 120 ./build/gemm_check f64 N N 32768 32768 256 5 144 -1 0 0 0 | grep -o "m=.*best=.*" || exit 1
echo "16384^3 with pad"
timeout -k 10 120 ./build/gemm_check f64 N N 16384 16384 16384 3 144 1 0 1 0 | grep -o "m=.*best=.*" || exit 1
} 2>&1 | tee gpurun_out/pad.log
