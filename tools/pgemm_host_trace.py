import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import eigen_amd
n = 8192
a = eigen_amd.fill_uniform_host(np.float64, n, n, 1)
b = eigen_amd.fill_uniform_host(np.float64, n, n, 2)
c0 = eigen_amd.fill_uniform_host(np.float64, n, n, 3)
for knobs in (dict(EIGAMD_MULTI_GPU="0"), dict(EIGAMD_MULTI_GPU_VIRTUAL="4", EIGAMD_PGEMM_TRACE="1")):
    os.environ.update(knobs)
    for i in range(3):
        c = c0.copy(order="F")
        t0 = time.perf_counter()
        eigen_amd.f77_gemm("N", "N", n, n, n, 1.0, a, n, b, n, 1.0, c, n)
        print(knobs, f"{(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
