#!/bin/bash
# LU 32768^2 with the driver's thresholds varied through the environment (one line per setting; A=0 = the defaults).
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python tools/lu_time.py 32768 2>&1 | grep LU | cut -c1-40; }
for rep in 1 2; do
run A=0
run EIGAMD_LU_WIDE_T2=14336
run EIGAMD_LU_WIDE_T2=14336 EIGAMD_LU_SLICE_MIN=20480
run EIGAMD_LU_WIDE_T2=14336 EIGAMD_LU_SLICE_MIN=20480 EIGAMD_LU_SLICE=1280
run EIGAMD_LU_WIDE_T2=12288 EIGAMD_LU_WIDE_T4=16384 EIGAMD_LU_SLICE_MIN=20480
run EIGAMD_LU_WIDE_T2=10240 EIGAMD_LU_WIDE_T4=16384
done
