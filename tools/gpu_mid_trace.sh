#!/bin/bash
# Kernel trace of device-resident square products of few rounds of tiles: how a product is cut into launches.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/mid_trace
rm -rf $OUT; mkdir -p $OUT
for n in ${@:-2560 3072}; do
  ./build/gemm_check f64 N N $n $n $n 5 0 1 0 1 0 | grep -o "m=.*best=.*"
  timeout -k 10 120 rocprofv3 --kernel-trace -d $OUT/t$n -o t -- ./build/gemm_check f64 N N $n $n $n 3 0 1 0 1 0 > $OUT/t$n.log 2>&1
  echo "n=$n profiler exit status $?"
  python3 - $(find $OUT/t$n -name "*.db") <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select start, end, name, grid_x/workgroup_x from kernels order by start").fetchall()
t0 = rows[0][0]
for s, e, n, g in rows[-8:]:
    print(f"  +{(s-t0)/1e3:9.1f} us  {(e-s)/1e3:8.1f} us  wgs {g:5d}  {n.split('(')[0][-60:]}")
PY
done
