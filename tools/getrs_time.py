#!/usr/bin/env python3
"""Times eigamd_getrs_dev behind eigamd_getrf_dev: n x n factors, nrhs right-hand sides (2 n^2 nrhs flop)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eigen_amd
from eigen_amd import _lib
from eigen_amd.device import DeviceMatrix
from eigen_amd import lu as ealu
for n, nrhs in ((8192, 8192), (16384, 4096), (32768, 2048), (32768, 32768)):
    a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(21)
    transp, nb, fz = ealu.getrf_dev(a)
    b = DeviceMatrix(n, nrhs, eigen_amd.F64)
    for op, name in ((_lib.OP_N, "N"), (_lib.OP_T, "T")):
        best = 1e30
        for _ in range(2):
            b.fill_uniform(22)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ealu.getrs_dev(op, a, transp, b)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"dgetrs {name} n={n} nrhs={nrhs}: {best:.2f} ms  {2.0 * n * n * nrhs / best / 1e9:.2f} TFLOP/s", flush=True)
    del a, b
