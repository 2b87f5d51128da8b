#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/exit_fault2
rm -rf $OUT; mkdir -p $OUT
for what in gemm trsm lu; do
  timeout -k 10 120 rocprofv3 --kernel-trace -d $OUT/$what -o t -- python3 tools/exit_fault_probe.py $what > $OUT/$what.log 2>&1
  echo "$what: exit status $?" | tee -a $OUT/summary.txt
done
grep -h "MAP\|SIGSEGV\|    @" $OUT/lu.log | head -80 >> $OUT/summary.txt
tail -60 $OUT/summary.txt
