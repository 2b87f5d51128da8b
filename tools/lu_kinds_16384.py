import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, eigen_amd
from eigen_amd.device import DeviceMatrix
from eigen_amd import lu as ealu
for kind, name, f in ((eigen_amd.C64, "z", 4.0), (eigen_amd.C32, "c", 4.0), (eigen_amd.F32, "s", 1.0)):
    for n in (14336, 16384):
        a = DeviceMatrix(n, n, kind)
        best = 1e30
        for rep in range(2):
            a.fill_uniform(77); torch.cuda.synchronize(); t0 = time.perf_counter()
            ealu.getrf_dev(a); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"{name}getrf n={n}: {best*1e3:.1f} ms  {f*2/3*n**3/best/1e12:.2f} TFLOP/s", flush=True)
