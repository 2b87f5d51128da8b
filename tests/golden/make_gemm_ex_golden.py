#!/usr/bin/env python3
"""Generates tests/golden/gemm_ex_golden.npz from the reference itself: outputs of Eigen's
operator* (ref_opstar_ex in oracle/ref_harness.cpp, AVX2+FMA build) for the two forms of
general_matrix_matrix_product that no F77 routine expresses -- one real and one complex
operand (test/mixingtypes.cpp), and a destination with an inner stride
(test/product.h:256-269).

Run in the build container (needs oracle/_ref):  python tests/golden/make_gemm_ex_golden.py

Inputs are not stored: they are regenerated from (seed, shape) by the library's counter-based
generator.  Stored per case: the parameters and the m x n result (a strided sample of it for
the larger cases).
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

SEED_A, SEED_B, SEED_C = 111, 222, 333
CODE = {"N": 0, "T": 1, "C": 2, "J": 3}


def case_list():
    cases = []
    sizes = [(1, 1, 1), (3, 5, 2), (17, 13, 9), (64, 64, 64), (65, 63, 129), (130, 70, 257), (300, 20, 130)]
    scal = [(1.0 + 0j, 1.0 + 0j), (0.7 - 0.9j, 1.3 - 1.1j), (-1.0 + 0j, 1.0 + 0j), (0.7 + 0j, 0j), (-2.0 + 0j, 0.5 + 0j)]
    # mixed: kind = complex result kind; which = operand that is real
    for kind in (2, 3):
        for which in "ab":
            for si, (m, n, k) in enumerate(sizes):
                for io, (oa, ob) in enumerate([("N", "N"), ("N", "T"), ("T", "N"), ("C", "T"), ("J", "N"), ("N", "J"),
                                               ("C", "C"), ("T", "T")]):
                    if (si + io) % 2 and m * n > 400:
                        continue
                    alpha, beta = scal[(si + io) % len(scal)]
                    cases.append(dict(kind=kind, real=which, opa=oa, opb=ob, m=m, n=n, k=k, alpha=alpha, beta=beta,
                                      pad=(si + io) % 3, c_inc=1 if (si + io) % 4 else 2))
    # same-kind products into a strided destination
    for kind in range(4):
        for si, (m, n, k) in enumerate(sizes):
            for io, (oa, ob) in enumerate([("N", "N"), ("T", "N"), ("N", "C")]):
                alpha, beta = scal[(si + io + 1) % len(scal)]
                if kind < 2:
                    alpha, beta = alpha.real, beta.real
                cases.append(dict(kind=kind, real="", opa=oa, opb=ob, m=m, n=n, k=k, alpha=alpha, beta=beta,
                                  pad=(si + io) % 3, c_inc=2 + (si + io) % 2))
    return cases


def operands(c):
    """a, b, c0 as views plus their parent buffers; C is an (ldc x n) buffer whose element
    (i, j) of the product lives at row i * c_inc."""
    dt = ol.DTYPES[c["kind"]]
    rdt = ol.DTYPES[c["kind"] - 2] if c["kind"] >= 2 else dt
    m, n, k, pad, inc = c["m"], c["n"], c["k"], c["pad"], c["c_inc"]
    ar, ac = (m, k) if c["opa"] in "NJ" else (k, m)
    br, bc = (k, n) if c["opb"] in "NJ" else (n, k)
    a = eigen_amd.fill_uniform_host(rdt if c["real"] == "a" else dt, ar, ac, SEED_A, ld=ar + pad)
    b = eigen_amd.fill_uniform_host(rdt if c["real"] == "b" else dt, br, bc, SEED_B, ld=br + pad)
    ldc = (m - 1) * inc + 1 + pad
    cbuf = eigen_amd.fill_uniform_host(dt, ldc, n, SEED_C, ld=ldc)
    return a, b, cbuf, ldc


def parent(view):
    return view.base if view.base is not None else view


def main():
    ref = ol.load_ref("v3")
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    out, meta = {}, []
    for idx, c in enumerate(case_list()):
        a, b, cbuf, ldc = operands(c)
        m, n, k, inc = c["m"], c["n"], c["k"], c["c_inc"]
        dt = ol.DTYPES[c["kind"]]
        got = np.array(cbuf, order="F", copy=True)
        al, be = np.array([c["alpha"]], dtype=dt), np.array([c["beta"]], dtype=dt)
        rc = ref.ref_opstar_ex(c["kind"], int(c["real"] == "a"), int(c["real"] == "b"), CODE[c["opa"]], CODE[c["opb"]], m, n,
                               k, ol._vp(al), ol._vp(parent(a)), a.shape[0] + c["pad"], ol._vp(parent(b)),
                               b.shape[0] + c["pad"], ol._vp(be), ol._vp(got), inc, ldc)
        assert rc == 0
        # everything outside the strided window must be untouched by the reference too
        mask = np.ones(got.shape, dtype=bool)
        mask[0:(m - 1) * inc + 1:inc, :] = False
        assert np.array_equal(got[mask], cbuf[mask])
        res = got[0:(m - 1) * inc + 1:inc, :]
        stride = 1 if m * n <= 1200 else 5
        out[f"res_{idx}"] = res[::stride, ::stride].copy()
        cc = dict(c)
        cc["alpha"] = [complex(c["alpha"]).real, complex(c["alpha"]).imag]
        cc["beta"] = [complex(c["beta"]).real, complex(c["beta"]).imag]
        cc["stride"] = stride
        meta.append(cc)
    out["meta_json"] = np.frombuffer(json.dumps(dict(seeds=[SEED_A, SEED_B, SEED_C], cases=meta,
                                                     simd=ref.ref_simd().decode())).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "gemm_ex_golden.npz")
    np.savez_compressed(path, **out)
    print(f"{len(meta)} cases -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
