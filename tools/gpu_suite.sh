#!/bin/bash
# Device pass: GPU test suite, smoke, bench, kernel-trace profile of the bench.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/pytest_gpu.log || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log || exit 1
timeout -k 10 300 python bench.py 2>&1 | tee gpurun_out/bench.log || exit 1
