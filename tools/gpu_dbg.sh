#!/bin/bash
# Concurrency check: several processes multiply on one GPU at the same time (build/gemm_check),
# then the multi-rank test file, for the shipped library and for a variant given as $1.
set -o pipefail
mkdir -p gpurun_out
variant=${1:-base}
{
for lib in shipped $variant; do
  if [ $lib = shipped ]; then export LD_LIBRARY_PATH=; else export LD_LIBRARY_PATH=$PWD/build/variants/$lib; fi
  echo "=== $lib: 4 concurrent processes, f64 / f32"
  for dt in f64 f32; do
    for i in 1 2 3 4; do
      ( for r in 1 2 3 4 5 6; do timeout -k 10 60 ./build/gemm_check $dt N N 1024 512 128 20 0 1 0 1 0 | grep -o "max_ratio=[^ ]*.*\(PASS\|FAIL\)" | sed -e 's/best.*TFLOP.s//' ; done | sort | uniq -c | sed -e "s/^/$dt p$i /" ) &
    done
    wait
  done
done
} 2>&1 | tee gpurun_out/dbg.log
