#!/bin/bash
# MFMA utilisation and clock of kernel variants:  name:dtype:size ...
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_util
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 60 ./build/mfma_peak | tee $OUT/mfma_peak.log
for v in "$@"; do
  IFS=: read name dt n <<< "$v"
  export LD_LIBRARY_PATH=$PWD/build/variants/$name
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
    tag=${name}_${dt}_$(echo $pass | cut -c1-8)
    timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/$tag -o r -- ./build/gemm_check $dt N N $n $n $n 1 > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/$tag.log; }
  done
  echo "$name $dt: $(grep -o 'best=.*TFLOP/s' $OUT/$tag.log)"
done
python3 tools/rocpd_summary.py $OUT/*/r_results.db | grep -E "^==|gemm_kernel" | grep -vE "^\s+[0-9]+\s" 
