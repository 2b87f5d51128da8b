#!/usr/bin/env python3
"""Generates tests/golden/gemm_golden.npz from the reference itself.

Run in the build container (needs oracle/_ref, i.e. /root/reference at build
time):  python tests/golden/make_golden.py

Inputs are NOT stored: they are regenerated from (seed, shape) by the library's
counter-based generator (eigen_amd.fill_uniform_host, host-only code).  Stored
per case: the parameters and the outputs of Eigen's F77 entry point (?gemm_,
blas/level3_impl.h:12-76) and of Eigen's operator* path (ref_opstar in
oracle/ref_harness.cpp), both from the AVX2+FMA build of the reference.  Large
cases keep a strided sample of C (every `stride`-th row and column).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
import eigen_amd  # noqa: E402

SEED_A, SEED_B, SEED_C = 101, 202, 303


def case_list():
    cases = []
    sizes = [(1, 1, 1), (3, 5, 2), (17, 13, 9), (31, 33, 20), (64, 64, 64), (65, 63, 129), (130, 70, 257),
             (257, 129, 100), (300, 20, 700)]
    for kind in range(4):
        ops = "NTC"
        alphas = [(1.0, 1.0), (0.7, 1.3), (-1.0, 1.0), (0.7, 0.0)] if kind < 2 else \
                 [(1.0 + 0j, 1.0 + 0j), (0.7 - 0.9j, 1.3 - 1.1j), (-1.0 + 0j, 1.0 + 0j), (0.7 - 0.9j, 0j)]
        for si, (m, n, k) in enumerate(sizes):
            for ia, oa in enumerate(ops):
                for ib, ob in enumerate(ops):
                    if kind < 2 and ("C" in (oa, ob)) and si % 3:
                        continue  # 'C' == 'T' for real kinds: keep a few
                    alpha, beta = alphas[(si + ia + 2 * ib) % len(alphas)]
                    pad = (si + ia) % 3
                    cases.append(dict(kind=kind, opa=oa, opb=ob, m=m, n=n, k=k, alpha=alpha, beta=beta, pad=pad))
        # conjugate-without-transpose forms reachable from operator* only
        if kind >= 2:
            for oa, ob in [("J", "N"), ("N", "J"), ("J", "C"), ("T", "J"), ("J", "J")]:
                cases.append(dict(kind=kind, opa=oa, opb=ob, m=33, n=29, k=41, alpha=0.7 - 0.9j, beta=1.3 - 1.1j, pad=1))
    return cases


def operands(c):
    dt = ol.DTYPES[c["kind"]]
    m, n, k, pad = c["m"], c["n"], c["k"], c["pad"]
    ar, ac = (m, k) if c["opa"] in "NJ" else (k, m)
    br, bc = (k, n) if c["opb"] in "NJ" else (n, k)
    a = eigen_amd.fill_uniform_host(dt, ar, ac, SEED_A, ld=ar + pad)
    b = eigen_amd.fill_uniform_host(dt, br, bc, SEED_B, ld=br + pad)
    c0 = eigen_amd.fill_uniform_host(dt, m, n, SEED_C, ld=m + pad)
    return a, b, c0


def full_buffer(view):
    """The ld x cols parent buffer of a fill_uniform_host view."""
    return view.base if view.base is not None else view


def main():
    ref = ol.load_ref("v3")
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    out = {}
    meta = []
    for idx, c in enumerate(case_list()):
        a, b, c0 = operands(c)
        m, n, k, pad = c["m"], c["n"], c["k"], c["pad"]
        lda, ldb, ldc = a.shape[0] + pad, b.shape[0] + pad, m + pad
        dt = ol.DTYPES[c["kind"]]
        c_op = np.array(full_buffer(c0), order="F", copy=True)
        code = {"N": 0, "T": 1, "C": 2, "J": 3}
        al, be = np.array([c["alpha"]], dtype=dt), np.array([c["beta"]], dtype=dt)
        ref.ref_opstar(c["kind"], code[c["opa"]], code[c["opb"]], m, n, k, ol._vp(al), ol._vp(full_buffer(a)), lda,
                       ol._vp(full_buffer(b)), ldb, ol._vp(be), ol._vp(c_op), ldc)
        stride = 1 if m * n <= 1200 else 7
        out[f"opstar_{idx}"] = c_op[:m:stride, :n:stride].copy()
        if "J" not in (c["opa"], c["opb"]):
            c_f77 = np.array(full_buffer(c0), order="F", copy=True)
            ol.call_f77(getattr(ref, ol.SUFFIX[c["kind"]] + "gemm_"), c["opa"], c["opb"], m, n, k, c["alpha"],
                        full_buffer(a), lda, full_buffer(b), ldb, c["beta"], c_f77, ldc)
            out[f"f77_{idx}"] = c_f77[:m:stride, :n:stride].copy()
        cc = dict(c)
        cc["alpha"] = [complex(c["alpha"]).real, complex(c["alpha"]).imag]
        cc["beta"] = [complex(c["beta"]).real, complex(c["beta"]).imag]
        cc["stride"] = stride
        meta.append(cc)
    out["meta_json"] = np.frombuffer(json.dumps(dict(seeds=[SEED_A, SEED_B, SEED_C], cases=meta,
                                                     simd=ref.ref_simd().decode())).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "gemm_golden.npz")
    np.savez_compressed(path, **out)
    print(f"{len(meta)} cases -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
