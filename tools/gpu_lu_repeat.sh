#!/bin/bash
# The LU parity tests several times in one process each, for the shipped library and for variants:
# tells a result that depends on the build from one that depends on the run.
set -o pipefail
mkdir -p gpurun_out
cp eigen_amd/lib/libeigen_amd_blas.so /tmp/shipped.so
{
for v in "$@"; do
  echo "=== $v"
  [ "$v" = ship ] && cp /tmp/shipped.so eigen_amd/lib/libeigen_amd_blas.so || cp build/variants/$v/libeigen_amd_blas.so eigen_amd/lib/libeigen_amd_blas.so
  timeout -k 10 400 python -u -c "
import pytest, sys
res = [int(pytest.main(['-q', 'tests/test_lu_gpu.py', '-m', 'gpu', '--timeout', '120', '--timeout-method', 'thread', '-p', 'no:cacheprovider'])) for _ in range(5)]
print('exit codes of 5 runs:', res)
"
done
cp /tmp/shipped.so eigen_amd/lib/libeigen_amd_blas.so
} 2>&1 | tee gpurun_out/lu_repeat.log
