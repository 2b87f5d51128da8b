#!/bin/bash
# Short- and long-reduction GEMM timing, GEMM / LU parity tests and LU timing with the library as built.
set -o pipefail
mkdir -p gpurun_out
{
for shape in "32768 32768 256" "16384 16384 256" "8192 8192 256" "16384 16384 1024" "8192 8192 2048" "16384 16384 16384"; do
  timeout -k 10 120 ./build/gemm_check f64 N N $shape 5 0 -1 0 1 0 | grep -o "m=.*best=.*" || exit 1
done
timeout -k 10 120 ./build/gemm_check f32 N N 16384 16384 512 5 0 -1 0 1 0 | grep -o "m=.*best=.*" || exit 1
timeout -k 10 120 ./build/gemm_check f32 N N 16384 16384 16384 3 0 1 0 1 0 | grep -o "m=.*best=.*" || exit 1
timeout -k 10 120 ./build/gemm_check c64 N N 8192 8192 512 5 0 -1 0 1 0 | grep -o "m=.*best=.*" || exit 1
timeout -k 10 600 python -u -m pytest tests/test_gemm_gpu.py tests/test_lu_gpu.py tests/test_syrk_gpu.py tests/test_trsm_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread || exit 1
timeout -k 10 200 python -u tools/lu_time.py 16384 32768 || exit 1
} 2>&1 | tee gpurun_out/epi.log
