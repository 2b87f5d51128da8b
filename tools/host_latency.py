#!/usr/bin/env python3
"""Host-to-host latency of dgemm_ (host pointers, C += A*B) for small and medium n, next to Eigen's
own CPU product on this host (oracle/_ref: the reference compiled where it lies; baseline only)."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as ol
import eigen_amd
from eigen_amd import _lib

ref = ol.load_ref()
sizes = [int(x) for x in sys.argv[1:]] or [32, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048]
print(f"{'n':>6} {'dgemm_ (GPU lib) us':>20} {'GFLOP/s':>9} {'Eigen 1 thread us':>18} {'Eigen all threads us':>21}")
for n in sizes:
    a = eigen_amd.fill_uniform_host(np.float64, n, n, 1)
    b = eigen_amd.fill_uniform_host(np.float64, n, n, 2)
    c = np.ones((n, n), order="F")
    ci = lambda v: ctypes.byref(ctypes.c_int(v))
    one = np.array([1.0])
    args = (b"N", b"N", ci(n), ci(n), ci(n), one.ctypes.data, a.ctypes.data, ci(n), b.ctypes.data, ci(n), one.ctypes.data,
            c.ctypes.data, ci(n))
    for _ in range(3):
        _lib.lib.dgemm_(*args)
    best = 1e9
    for _ in range(20):
        t0 = time.perf_counter(); _lib.lib.dgemm_(*args); best = min(best, time.perf_counter() - t0)
    cpu1 = cpuN = float("nan")
    if ref is not None:
        ref.ref_set_threads(1); cpu1 = ref.ref_bench_gemm(1, n, n, n, 10)
        ref.ref_set_threads(min(16, os.cpu_count() or 1)); cpuN = ref.ref_bench_gemm(1, n, n, n, 10)
    print(f"{n:6d} {best * 1e6:20.1f} {2.0 * n ** 3 / best / 1e9:9.1f} {cpu1 * 1e6:18.1f} {cpuN * 1e6:21.1f}", flush=True)
