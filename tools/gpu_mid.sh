#!/bin/bash
# Device-resident square products between 512 and 4096: how well launches of few tiles use the chip.
set -o pipefail
mkdir -p gpurun_out
{
for dt in ${DTYPES:-f64 f32}; do
for n in ${@:-512 1024 1536 2048 2560 3072 4096}; do
  timeout -k 10 120 ./build/gemm_check $dt N N $n $n $n 5 0 1 0 1 0 | grep -o "m=.*best=.*" || exit 1
done
done
} 2>&1 | tee gpurun_out/mid.log
