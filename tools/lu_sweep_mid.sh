#!/bin/bash
# LU at 16384 ... 32768 with the outer-step width thresholds varied (A=0 = the defaults).
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python tools/lu_time.py 16384 20480 24576 32768 2>&1 | grep LU | cut -c8-32 | tr '\n' ' '; echo; }
for rep in 1 2; do
run A=0
run EIGAMD_LU_WIDE_T2=12288
run EIGAMD_LU_WIDE_T2=12288 EIGAMD_LU_WIDE_T4=18432
run EIGAMD_LU_WIDE_T2=12288 EIGAMD_LU_WIDE_T4=16384
run EIGAMD_LU_WIDE_T2=12288 EIGAMD_LU_WIDE_T4=14336
done
