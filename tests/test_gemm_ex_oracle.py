"""Pins the oracle's restatement of the two product forms no F77 routine expresses -- one real
and one complex operand, and a destination with an inner stride -- against outputs of Eigen's
own operator* (tests/golden/gemm_ex_golden.npz) and, where oracle/_ref was built, against the
reference on fresh inputs.  Tolerance: the Netlib test ratio (see test_gemm_gpu.py) below 16."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from test_oracle import RATIO_LIMIT, opmat

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_EX = os.path.join(HERE, "golden", "gemm_ex_golden.npz")
CODE = {"N": 0, "T": 1, "C": 2, "J": 3}


def golden_ex():
    z = np.load(GOLDEN_EX)
    return z, json.loads(bytes(z["meta_json"]).decode())


def operands_ex(c, seeds):
    """Operands of a golden case (same construction as tests/golden/make_gemm_ex_golden.py)."""
    dt = ol.DTYPES[c["kind"]]
    rdt = ol.DTYPES[c["kind"] - 2] if c["kind"] >= 2 else dt
    m, n, k, pad, inc = c["m"], c["n"], c["k"], c["pad"], c["c_inc"]
    ar, ac = (m, k) if c["opa"] in "NJ" else (k, m)
    br, bc = (k, n) if c["opb"] in "NJ" else (n, k)
    a = eigen_amd.fill_uniform_host(rdt if c["real"] == "a" else dt, ar, ac, seeds[0], ld=ar + pad)
    b = eigen_amd.fill_uniform_host(rdt if c["real"] == "b" else dt, br, bc, seeds[1], ld=br + pad)
    ldc = (m - 1) * inc + 1 + pad
    cbuf = eigen_amd.fill_uniform_host(dt, ldc, n, seeds[2], ld=ldc)
    return a, b, cbuf, ldc


def parent(v):
    return v.base if v.base is not None else v


def window(cbuf, m, inc):
    return cbuf[0:(m - 1) * inc + 1:inc, :]


def ratio_ex(c, got, want, a, b, c0win):
    abs1 = lambda z: np.abs(z.real) + np.abs(z.imag)
    alpha, beta = complex(*c["alpha"]), complex(*c["beta"])
    g = abs1(alpha) * (abs1(opmat(a, c["opa"])).astype(np.float64) @ abs1(opmat(b, c["opb"])).astype(np.float64))
    g = g + abs1(beta) * abs1(c0win)
    s = c["stride"]
    g = np.where(g == 0, 1.0, g)[::s, ::s]
    return float((abs1(got.astype(np.complex128) - want.astype(np.complex128)) / (ol.EPS[c["kind"]] * g)).max())


def scalars(c):
    alpha, beta = complex(*c["alpha"]), complex(*c["beta"])
    return (alpha.real, beta.real) if c["kind"] < 2 else (alpha, beta)


def run_case(fn, c, seeds):
    """fn(opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, cbuf, c_inc, ldc) on a golden case;
    returns (result window sample, operands); checks that nothing outside the window changed."""
    a, b, c0, ldc = operands_ex(c, seeds)
    m, n, k, inc, pad = c["m"], c["n"], c["k"], c["c_inc"], c["pad"]
    alpha, beta = scalars(c)
    cbuf = np.array(c0, order="F", copy=True)
    fn(c["opa"], c["opb"], m, n, k, alpha, parent(a), a.shape[0] + pad, parent(b), b.shape[0] + pad, beta, cbuf, inc, ldc)
    mask = np.ones(cbuf.shape, dtype=bool)
    mask[0:(m - 1) * inc + 1:inc, :] = False
    assert np.array_equal(cbuf[mask], c0[mask]), ("wrote between / below the strided rows", c)
    s = c["stride"]
    return window(cbuf, m, inc)[::s, ::s], a, b, window(c0, m, inc)


def test_oracle_ex_vs_golden():
    z, meta = golden_ex()
    assert len(meta["cases"]) == 244
    worst = 0.0
    for idx, c in enumerate(meta["cases"]):
        got, a, b, c0w = run_case(ol.oracle_product_ex, c, meta["seeds"])
        r = ratio_ex(c, got, z[f"res_{idx}"], a, b, c0w)
        worst = max(worst, r)
        assert r < RATIO_LIMIT, (idx, c, r)
    assert worst > 0.0


ref = ol.load_ref()


@pytest.mark.skipif(ref is None, reason="oracle/_ref not built (needs the reference tree at build time)")
def test_oracle_ex_vs_reference_fresh_inputs():
    seeds = [5, 6, 7]
    scal = [(1.0 + 0j, 1.0 + 0j), (0.7 - 0.9j, 1.3 - 1.1j), (0.3 + 0j, 0j)]
    i = 0
    for kind in (2, 3):
        for real in "ab":
            for opa in "NTCJ":
                for opb in "NTCJ":
                    i += 1
                    alpha, beta = scal[i % 3]
                    c = dict(kind=kind, real=real, opa=opa, opb=opb, m=70 + i % 5, n=41, k=90 + i % 7,
                             alpha=[alpha.real, alpha.imag], beta=[beta.real, beta.imag], pad=i % 3, c_inc=1 + i % 3,
                             stride=1)
                    got, a, b, c0w = run_case(ol.oracle_product_ex, c, seeds)

                    def ref_fn(opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, cbuf, inc, ldc):
                        dt = cbuf.dtype
                        al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
                        assert ref.ref_opstar_ex(kind, int(real == "a"), int(real == "b"), CODE[opa], CODE[opb], m, n, k,
                                                 ol._vp(al), ol._vp(a), lda, ol._vp(b), ldb, ol._vp(be), ol._vp(cbuf), inc,
                                                 ldc) == 0
                    want, _, _, _ = run_case(ref_fn, c, seeds)
                    r = ratio_ex(c, got, want, a, b, c0w)
                    assert r < RATIO_LIMIT, (c, r)
