#!/bin/bash
# Which programs end with a fault under `rocprofv3 --kernel-trace` (after their own work is done)?
# One short run each; the exit status of every run is written down, nothing is retried.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/exit_fault
rm -rf $OUT; mkdir -p $OUT
run() {  # tag, command...
  tag=$1; shift
  timeout -k 10 120 rocprofv3 --kernel-trace -d $OUT/$tag -o t -- "$@" > $OUT/$tag.log 2>&1
  echo "$tag: exit status $? :: $*" | tee -a $OUT/summary.txt
}
run torch_only python3 -c "import torch; x = torch.ones(1024, device='cuda'); print(float(x.sum()))"
run lib_device_path ./build/gemm_check f64 N N 2048 2048 2048 2
run lib_host_path ./oracle/_ref/dropin_blas bench 2048
run torch_and_lib python3 tools/lu_time.py 2048
grep -h -A12 "SIGSEGV\|Segmentation" $OUT/*.log | head -60 >> $OUT/summary.txt
cat $OUT/summary.txt
