"""?gemm_ called from several host threads at once (Eigen products inside the caller's own parallel region;
the reference's ?gemm_ keeps no state between calls, blas/level3_impl.h:12-76): every thread's results
against the CPU oracle computed beforehand, to the Netlib test ratio (< 16, as in test_gemm_gpu.py).  Small
products take a context of their own per call, large ones share the staging pipeline behind a lock; the
mix below has both in flight together."""
import threading

import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from test_gemm_gpu import ratio

pytestmark = pytest.mark.gpu

CASES = [(0, 33, 47, 29, "N", "N"), (1, 64, 64, 64, "T", "N"), (1, 130, 70, 200, "N", "T"), (3, 50, 60, 70, "C", "N"),
         (2, 97, 31, 55, "N", "C"), (1, 256, 256, 256, "N", "N"), (0, 300, 200, 100, "T", "T"), (1, 900, 700, 650, "N", "N")]


def make_case(kind, m, n, k, ta, tb, seed):
    dt = ol.DTYPES[kind]
    a = eigen_amd.fill_uniform_host(dt, *((m, k) if ta == "N" else (k, m)), seed)
    b = eigen_amd.fill_uniform_host(dt, *((k, n) if tb == "N" else (n, k)), seed + 1)
    c = eigen_amd.fill_uniform_host(dt, m, n, seed + 2)
    alpha, beta = ((0.7 - 0.9j), (1.3 - 1.1j)) if kind >= 2 else (0.7, 1.3)
    want = c.copy(order="F")
    ol.call_f77(ol.oracle_gemm(kind), ta, tb, m, n, k, alpha, a, a.shape[0], b, b.shape[0], beta, want, m)
    return dict(kind=kind, m=m, n=n, k=k, ta=ta, tb=tb, a=a, b=b, c=c, alpha=alpha, beta=beta, want=want)


def op(x, t):
    return x if t == "N" else (x.T if t == "T" else x.conj().T)


def test_concurrent_host_calls():
    nthreads, rounds = 8, 6
    work = [[make_case(*CASES[(t + r) % len(CASES)], seed=100 * t + 10 * r) for r in range(rounds)] for t in range(nthreads)]
    results = [[None] * rounds for _ in range(nthreads)]
    start = threading.Barrier(nthreads)

    def run(t):
        start.wait()
        for r, w in enumerate(work[t]):
            got = w["c"].copy(order="F")
            eigen_amd.f77_gemm(w["ta"], w["tb"], w["m"], w["n"], w["k"], w["alpha"], w["a"], w["a"].shape[0], w["b"],
                               w["b"].shape[0], w["beta"], got, w["m"])
            results[t][r] = got

    threads = [threading.Thread(target=run, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for t in range(nthreads):
        for r, w in enumerate(work[t]):
            rr = ratio(w["kind"], results[t][r], w["want"], w["alpha"], op(w["a"], w["ta"]), op(w["b"], w["tb"]), w["beta"], w["c"])
            assert rr < 16.0, (t, r, CASES[(t + r) % len(CASES)], rr)
