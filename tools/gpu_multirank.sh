#!/bin/bash
# The multi-rank GPU test file, repeated, progress streamed; stops at the first failing run.
set -o pipefail
mkdir -p gpurun_out
export EIGAMD_TEST_VERBOSE=1
for r in $(seq ${1:-3}); do
  echo "=== run $r"
  timeout -k 10 500 python -u -m pytest tests/test_y_multirank_gpu.py -x -q -s --durations=0 --timeout 200 --timeout-method thread 2>&1 | grep --line-buffered -v "Gloo\|amdgpu.ids\|socket.cpp\|update" | tee -a gpurun_out/multirank.log | grep --line-buffered "passed\|failed\|Timeout\|repo/\|s call\|rank" || exit 1
done
