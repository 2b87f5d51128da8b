#!/bin/bash
# Device-resident products whose extents are not multiples of the tile: how much the bounds-checked
# edge launches cost.
set -o pipefail
mkdir -p gpurun_out
{
for dt in f64 f32; do
for shape in "4096 4096 4096" "4000 4000 4000" "4100 4100 4100" "8000 8000 8000" "128 16384 128" "16384 100 16384" "1000 1000 1000"; do
  timeout -k 10 120 ./build/gemm_check $dt N N $shape 5 0 1 0 1 0 | grep -o "m=.*best=.*" || exit 1
done
done
} 2>&1 | tee gpurun_out/odd.log
