#!/bin/bash
# LU and k = 256 GEMM timing with a variant build copied over the library (box-local copy only).
set -o pipefail
mkdir -p gpurun_out
cp eigen_amd/lib/libeigen_amd_blas.so /tmp/shipped.so
{
for v in "$@"; do
  cp build/variants/$v/libeigen_amd_blas.so eigen_amd/lib/libeigen_amd_blas.so
  echo "=== $v"
  LD_LIBRARY_PATH=$PWD/build/variants/$v timeout -k 10 120 ./build/gemm_check f64 N N 32768 32768 256 5 0 -1 0 1 0 | grep -o "m=.*best=.*" || exit 1
  LD_LIBRARY_PATH=$PWD/build/variants/$v timeout -k 10 120 ./build/gemm_check f64 N N 16384 16384 256 5 0 -1 0 1 0 | grep -o "m=.*best=.*" || exit 1
  timeout -k 10 200 python -u tools/lu_time.py 16384 32768 || exit 1
done
cp /tmp/shipped.so eigen_amd/lib/libeigen_amd_blas.so
} 2>&1 | tee gpurun_out/lu_variant.log
