#!/bin/bash
# First device pass: MFMA ceiling, small parity cases, then the headline sizes.
set -e
mkdir -p gpurun_out
{
timeout -k 10 60 ./build/mfma_peak
for dt in f64 f32 c64 c32; do
  for ops in "N N" "T N" "N T" "T T"; do
    timeout -k 10 60 ./build/gemm_check $dt $ops 128 128 64 1 0 || exit 1
    timeout -k 10 60 ./build/gemm_check $dt $ops 257 131 77 1 1 0.7 0 1.3 0 || exit 1
  done
done
timeout -k 10 60 ./build/gemm_check c64 C N 200 150 90 1 3 0.7 -0.9 1.3 -1.1
timeout -k 10 60 ./build/gemm_check c64 N C 200 150 90 1 3 0.7 -0.9 1.3 -1.1
timeout -k 10 60 ./build/gemm_check c64 C C 200 150 90 1 3 0.7 -0.9 0 0
timeout -k 10 60 ./build/gemm_check c32 C J 200 150 90 1 3 0.7 -0.9 1.3 -1.1
timeout -k 10 60 ./build/gemm_check c32 J C 200 150 90 1 3 0.7 -0.9 1.3 -1.1
timeout -k 10 120 ./build/gemm_check f64 N N 4096 4096 4096 3
timeout -k 10 120 ./build/gemm_check f64 N N 16384 16384 16384 3
timeout -k 10 120 ./build/gemm_check f32 N N 16384 16384 16384 3
timeout -k 10 120 ./build/gemm_check c64 N N 8192 8192 8192 3
timeout -k 10 120 ./build/gemm_check c32 N N 8192 8192 8192 3
} 2>&1 | tee gpurun_out/first.log
