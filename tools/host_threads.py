#!/usr/bin/env python3
"""Host-to-host dgemm_ calls per second from 1, 2, 4 and 8 threads calling at the same time (n = 64, 256)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eigen_amd

def rate(n, nthreads, calls=300):
    ops = []
    for t in range(nthreads):
        a = eigen_amd.fill_uniform_host(np.float64, n, n, 1 + t)
        b = eigen_amd.fill_uniform_host(np.float64, n, n, 2 + t)
        c = eigen_amd.fill_uniform_host(np.float64, n, n, 3 + t)
        ops.append((a, b, c))
    for a, b, c in ops:
        eigen_amd.f77_gemm("N", "N", n, n, n, 1.0, a, n, b, n, 0.5, c, n)
    start = threading.Barrier(nthreads + 1)
    def run(t):
        a, b, c = ops[t]
        start.wait()
        for _ in range(calls):
            eigen_amd.f77_gemm("N", "N", n, n, n, 1.0, a, n, b, n, 0.5, c, n)
    th = [threading.Thread(target=run, args=(t,)) for t in range(nthreads)]
    for x in th: x.start()
    start.wait()
    t0 = time.perf_counter()
    for x in th: x.join()
    return nthreads * calls / (time.perf_counter() - t0)

for n in (64, 256):
    print(f"dgemm_ {n}^3 host to host, calls per second: " + "  ".join(f"{k} threads {rate(n, k):9.0f}" for k in (1, 2, 4, 8)), flush=True)
