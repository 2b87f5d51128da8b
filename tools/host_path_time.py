#!/usr/bin/env python3
"""Wall-clock rate of the host-pointer entry (dgemm_ on pageable numpy arrays): PCIe and
staging included."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eigen_amd
from eigen_amd import blas
for n in [int(x) for x in (sys.argv[1:] or ["4096", "8192", "16384"])]:
    a = eigen_amd.fill_uniform_host(np.float64, n, n, 1)
    b = eigen_amd.fill_uniform_host(np.float64, n, n, 2)
    c = eigen_amd.fill_uniform_host(np.float64, n, n, 3)
    best = 1e30
    for rep in range(3):
        t0 = time.perf_counter()
        blas.gemm(1.0, a, b, 1.0, c)
        best = min(best, time.perf_counter() - t0)
    print(f"dgemm_ host-to-host n={n}: {best*1e3:.1f} ms  {2*n**3/best/1e12:.2f} TFLOP/s  "
          f"({4*n*n*8/best/1e9:.1f} GB/s over PCIe both ways)", flush=True)
