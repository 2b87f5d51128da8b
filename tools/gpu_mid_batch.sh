#!/bin/bash
# Mid-size products timed alone and in batches of back-to-back launches.
set -o pipefail
mkdir -p gpurun_out
{
for dt in ${DTYPES:-f64 f32}; do
for b in 1 20; do
for n in ${@:-1024 1536 2048 2560 2816 3072 3328 3584 4096 4608 5120}; do
  echo -n "batch $b: "
  GEMM_CHECK_BATCH=$b timeout -k 10 120 ./build/gemm_check $dt N N $n $n $n 5 0 1 0 1 0 | grep -o "m=.*best=.*" || exit 1
done
done
done
} 2>&1 | tee gpurun_out/mid_batch.log
