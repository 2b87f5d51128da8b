#!/bin/bash
# Short-reduction GEMM (the k = 256 trailing update of the LU) with variant builds of the library.
set -o pipefail
mkdir -p gpurun_out
{
for v in "$@"; do
  echo "=== $v"
  for shape in "32768 32768 256" "16384 16384 256" "8192 8192 256" "16384 16384 1024" "16384 16384 16384"; do
    LD_LIBRARY_PATH=$PWD/build/variants/$v timeout -k 10 120 ./build/gemm_check f64 N N $shape 5 0 -1 0 1 0 | grep -o "m=.*best=.*" || exit 1
  done
done
} 2>&1 | tee gpurun_out/shortk.log
