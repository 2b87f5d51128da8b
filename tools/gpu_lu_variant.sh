#!/bin/bash
# LU parity test and timing with variant builds copied over the library (box-local copy only).
set -o pipefail
mkdir -p gpurun_out
cp eigen_amd/lib/libeigen_amd_blas.so /tmp/shipped.so
{
for v in "$@"; do
  echo "=== $v"
  [ "$v" = ship ] && cp /tmp/shipped.so eigen_amd/lib/libeigen_amd_blas.so || cp build/variants/$v/libeigen_amd_blas.so eigen_amd/lib/libeigen_amd_blas.so
  timeout -k 10 300 python -u -m pytest tests/test_lu_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread || exit 1
  timeout -k 10 200 python -u tools/lu_time.py 8192 16384 32768 || exit 1
done
cp /tmp/shipped.so eigen_amd/lib/libeigen_amd_blas.so
} 2>&1 | tee gpurun_out/lu_variant.log
