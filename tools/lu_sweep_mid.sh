#!/bin/bash
# LU at 4096 ... 16384 with the panel chain's thresholds varied (A=0 = the defaults).
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python tools/lu_time.py 4096 8192 12288 16384 2>&1 | grep LU | cut -c8-32 | tr '\n' ' '; echo; }
for rep in 1 2; do
run A=0
for v in 2304 8192 16384; do run EIGAMD_LU_FUSED_ROWOPS_MAX=$v; done
for v in 4096 8192 16384; do run EIGAMD_LU_FAR_RIDES_MAX=$v; done
for v in 4096 16384; do run EIGAMD_LU_SPLIT_MIN=$v; done
run EIGAMD_LU_RAMP=0
done
