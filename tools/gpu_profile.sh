#!/bin/bash
# rocprofv3 passes for the headline kernel.  Each pass is its own run; counter passes
# use --pmc only (never with a trace domain).
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_available.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline > $OUT/trace_bench.log 2>&1 || { tail -5 $OUT/trace_bench.log; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/pmc_$tag -o f64 -- ./build/gemm_check f64 N N 16384 16384 16384 2 > $OUT/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/pmc_$tag.log; }
done
find $OUT -name "*.csv" | head -50
