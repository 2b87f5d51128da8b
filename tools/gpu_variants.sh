#!/bin/bash
# A/B of tile-shape variants (build/variants/*) with build/gemm_check:  name:dtype:size
set -o pipefail
mkdir -p gpurun_out
{
for v in "$@"; do
  IFS=: read name dt n <<< "$v"
  export LD_LIBRARY_PATH=$PWD/build/variants/$name
  echo "=== $name ($dt)"
  timeout -k 10 60 ./build/gemm_check $dt N N 257 131 77 1 1 0.7 0 1.3 0 | grep -o "max_ratio.*" || exit 1
  timeout -k 10 60 ./build/gemm_check $dt C T 300 517 211 1 0 | grep -o "max_ratio.*" || exit 1
  timeout -k 10 120 ./build/gemm_check $dt N N $n $n $n 3 | grep -o "m=.*best=.*" || exit 1
  timeout -k 10 120 ./build/gemm_check $dt N N 4096 4096 4096 5 | grep -o "m=4096.*best=.*" || exit 1
done
} 2>&1 | tee gpurun_out/variants.log
