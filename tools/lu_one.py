#!/usr/bin/env python3
"""One device LU: lu_one.py kind n   (kind f32|f64|c32|c64); prints the time of the second run or the error."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, eigen_amd
from eigen_amd.device import DeviceMatrix
from eigen_amd import lu as ealu
kind = {"f32": eigen_amd.F32, "f64": eigen_amd.F64, "c32": eigen_amd.C32, "c64": eigen_amd.C64}[sys.argv[1]]
n = int(sys.argv[2])
a = DeviceMatrix(n, n, kind)
best = 1e30
for rep in range(2):
    a.fill_uniform(77); torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        transp, nb, fz = ealu.getrf_dev(a)
    except Exception as e:
        print(f"{sys.argv[1]} n={n}: {type(e).__name__}: {str(e)[:120]}", flush=True)
        sys.exit(1)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
t = transp.cpu().numpy()
ok = fz == -1 and bool((t >= np.arange(n)).all() and (t < n).all())
f = 8.0 if kind >= 2 else 2.0
print(f"{sys.argv[1]} getrf n={n}: {best*1e3:.1f} ms  {f/3*n**3/best/1e12:.2f} TFLOP/s  structure ok {ok}", flush=True)
