#!/usr/bin/env python3
"""Times the LU-shaped trailing update C -= A*B (m = n, short k) on device-resident operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eigen_amd
from eigen_amd.device import DeviceMatrix, gemm_dev
ld = 32768
big = DeviceMatrix(ld, ld, eigen_amd.F64).fill_uniform(1)
for n in [int(x) for x in (sys.argv[1:] or ["31744", "16384", "8192"])]:
    for k in (256, 512, 1024, 2048):
        a = big.block(ld - n, 0, n, k)
        b = big.block(0, ld - n, k, n)
        c = big.block(ld - n, ld - n, n, n)
        gemm_dev("N", "N", -1.0, a, b, 1.0, c)
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gemm_dev("N", "N", -1.0, a, b, 1.0, c); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"n={n} k={k}: {best:.3f} ms  {2.0*n*n*k/best/1e9:.1f} TFLOP/s", flush=True)
