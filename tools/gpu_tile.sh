#!/bin/bash
# Products with the large tile (1), the small tile (2) and the launcher's choice (0), batches of 20 launches.
# DT=f64 (default), f32, c64 or c32.
set -o pipefail
mkdir -p gpurun_out
{
DT=${DT:-f64}
VAR=EIGAMD_$(echo $DT | tr a-z A-Z)_TILE
for t in 1 2 0; do
for s in "512 512 512" "1024 1024 1024" "1536 1536 1536" "2048 2048 2048" "2560 2560 2560" "3072 3072 3072" "4096 4096 4096" "256 1024 256" "256 8192 256" "8192 256 256" "2048 2048 256" "8192 8192 256" "512 512 4096" "1024 1024 8192"; do
  set -- $s
  echo -n "tile $t: "
  env $VAR=$t GEMM_CHECK_BATCH=20 timeout -k 10 120 ./build/gemm_check $DT N N $1 $2 $3 5 0 1 0 1 0 | grep -o "m=.*best=.*" || exit 1
done
done
} 2>&1 | tee gpurun_out/tile_$DT.log
