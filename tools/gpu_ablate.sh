#!/bin/bash
# timing-only comparison of ablated builds (results of ablated builds are wrong by design)
for v in "$@"; do
  IFS=: read name dt n <<< "$v"
  export LD_LIBRARY_PATH=$PWD/build/variants/$name
  echo "=== $name ($dt) $(timeout -k 10 120 ./build/gemm_check $dt N N $n $n $n 3 | grep -o 'best=.*TFLOP/s')"
done
