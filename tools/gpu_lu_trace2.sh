#!/bin/bash
# Kernel trace of the device LU at one size, summary + per-queue timeline.
set -o pipefail
export TMPDIR=/tmp
n=${1:-32768}
OUT=$PWD/gpurun_out/prof_lu
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o lu -- python3 tools/lu_time.py $n > $OUT/trace.log 2>&1
echo "profiler exit status $?" | tee -a $OUT/trace.log
grep "LU f64" $OUT/trace.log
db=$(find $OUT/trace -name "*.db")
python3 tools/rocpd_summary.py $db | head -24 | tee $OUT/summary.txt
python3 tools/lu_timeline.py $db 50 | tee -a $OUT/summary.txt
