#!/usr/bin/env python3
"""Times the device-resident LU (eigamd_getrf_dev) at a few sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eigen_amd
from eigen_amd.device import DeviceMatrix
from eigen_amd import lu as ealu
for n in [int(x) for x in (sys.argv[1:] or ["4096", "8192", "16384", "32768"])]:
    a = DeviceMatrix(n, n, eigen_amd.F64)
    best = 1e30
    for rep in range(2):
        a.fill_uniform(77)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        transp, nb, fz = ealu.getrf_dev(a)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"LU f64 n={n}: {best*1e3:.1f} ms  {2/3*n**3/best/1e12:.2f} TFLOP/s  first_zero={fz} swaps={nb}", flush=True)
