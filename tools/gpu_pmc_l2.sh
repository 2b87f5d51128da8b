#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_l2
rm -rf $OUT; mkdir -p $OUT
for cfg in "8 1 0" "8 1 24000" "4 1 24000" "16 1 24000"; do
  set -- $cfg
  export EIGAMD_GROUP_M=$1 EIGAMD_XCD_REMAP=$2 EIGAMD_EXTRA_LDS=$3
  for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=gm$1_x$2_l$3_$(echo $pass | tr ' ' '_' | cut -c1-12)
    timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/$tag -o r -- ./build/gemm_check f64 N N 16384 16384 16384 1 > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/$tag.log; }
  done
  echo "cfg gm=$1 xcd=$2 lds=$3: $(grep -o 'best=.*TFLOP/s' $OUT/$tag.log)"
done
python3 tools/rocpd_summary.py $OUT/*/r_results.db | grep -E "^==|   (FETCH|TCC).*gemm_kernel" 
