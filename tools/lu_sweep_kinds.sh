run() { echo -n "$* : "; env "$@" timeout -k 10 200 python tools/lu_kinds_time.py 2>&1 | grep "n=8192\|n=4096" | cut -c1-26 | tr '\n' ' '; echo; }
for rep in 1 2; do
run A=0
run EIGAMD_LU_PANEL_CUS=96
run EIGAMD_LU_PANEL_CUS=128
run EIGAMD_LU_FAR_RIDES_MAX=6144
run EIGAMD_LU_FAR_RIDES_MAX=16384
run EIGAMD_LU_FUSED_ROWOPS_MAX=2304
run EIGAMD_LU_FUSED_ROWOPS_MAX=8192
done
