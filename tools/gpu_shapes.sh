#!/bin/bash
# Device-resident products of a list of shapes, one line each:  "kind m n k" ...
# Without arguments: the square sizes between 512 and 4096 (launches of few tiles) and the sizes whose last
# round of tiles is mostly empty, for the two real kinds.  Environment switches of the library
# (EIGAMD_SPLITK=0, EIGAMD_SMALL_TILE=0, ...) pass through, which is how a rule is compared with its absence.
set -o pipefail
mkdir -p gpurun_out
if [ $# -eq 0 ]; then
  set --
  for dt in f64 f32; do for n in 512 1024 1536 2048 2560 3072 4096 4608 5120 6144 8192; do set -- "$@" "$dt $n $n $n"; done; done
fi
{
for shape in "$@"; do
  read dt m n k <<< "$shape"
  timeout -k 10 120 ./build/gemm_check $dt N N $m $n $k 5 0 1 0 1 0 | grep -o "m=.*best=.*" | sed "s/^/$dt /; s/max_ratio.*samples//" || exit 1
done
} 2>&1 | tee gpurun_out/shapes.log
