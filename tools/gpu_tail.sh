#!/bin/bash
# Device-resident products of a few rounds of tiles plus a mostly empty last round.
set -o pipefail
mkdir -p gpurun_out
{
for shape in "f64 5120 5120 5120" "f64 4608 4608 4608" "f64 6144 6144 6144" "f64 3072 3072 3072" "c64 3072 3072 3072" "f32 5120 5120 5120" "f32 6400 6400 6400" "c32 3328 3328 3328" "f64 8192 8192 8192"; do
  set -- $shape
  timeout -k 10 120 ./build/gemm_check $1 N N $2 $3 $4 5 0 1 0 1 0 | grep -o "m=.*best=.*" | sed "s/^/$1 /; s/max_ratio.*samples//" || exit 1
done
} 2>&1 | tee gpurun_out/tail.log
