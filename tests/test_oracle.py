"""Pins the CPU oracle (oracle/gemm_oracle.c) before anything is compared to it:
the Netlib level-3 known-answer driver with the reference's own parameter files,
golden outputs of Eigen's ?gemm_ and operator*, and -- where oracle/_ref was
built -- the reference itself, bit for bit against its scalar build."""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")
GOLDEN = os.path.join(HERE, "golden", "gemm_golden.npz")
# Pass mark of the Netlib testers: test ratio |c - c_ref| / (eps * gauge) below the
# THRESHOLD line of blas/testing/?blat3.dat.
RATIO_LIMIT = 16.0


def golden():
    z = np.load(GOLDEN)
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


def golden_operands(case, seeds):
    dt = ol.DTYPES[case["kind"]]
    m, n, k, pad = case["m"], case["n"], case["k"], case["pad"]
    ar, ac = (m, k) if case["opa"] in "NJ" else (k, m)
    br, bc = (k, n) if case["opb"] in "NJ" else (n, k)
    a = eigen_amd.fill_uniform_host(dt, ar, ac, seeds[0], ld=ar + pad)
    b = eigen_amd.fill_uniform_host(dt, br, bc, seeds[1], ld=br + pad)
    c = eigen_amd.fill_uniform_host(dt, m, n, seeds[2], ld=m + pad)
    return a, b, c


def opmat(x, op):
    return {"N": x, "T": x.T, "C": x.conj().T, "J": x.conj()}[op]


def case_ratio(case, got, want, a, b, c0):
    abs1 = (lambda z: np.abs(z.real) + np.abs(z.imag)) if case["kind"] >= 2 else np.abs
    alpha, beta = complex(*case["alpha"]), complex(*case["beta"])
    s = case["stride"]
    g = (abs(alpha.real) + abs(alpha.imag)) * (abs1(opmat(a, case["opa"])).astype(np.float64)
                                              @ abs1(opmat(b, case["opb"])).astype(np.float64))
    g = g + (abs(beta.real) + abs(beta.imag)) * abs1(c0)
    g = g[::s, ::s]
    g = np.where(g == 0, 1.0, g)
    d = got.astype(np.complex128) - want.astype(np.complex128)
    return float(((np.abs(d.real) + np.abs(d.imag)) / (ol.EPS[case["kind"]] * g)).max())


@pytest.mark.parametrize("kind", range(4))
def test_netlib_known_answers_oracle(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    rep = ol.run_netlib_chk1(kind, ol.oracle_gemm(kind), p)
    assert rep.ncalls == 17496  # the call count the Fortran ?CHK1 reports for GEMM
    assert not rep.fatal and rep.nfail_same == 0 and rep.nfail_ratio == 0
    assert rep.errmax < p["thresh"] == RATIO_LIMIT


@pytest.mark.parametrize("kind", range(4))
def test_netlib_error_exits_oracle(kind):
    assert ol.run_netlib_chke(kind, ol.oracle_gemm(kind)) == 0


def test_oracle_vs_golden():
    z, meta = golden()
    worst = 0.0
    for idx, case in enumerate(meta["cases"]):
        a, b, c0 = golden_operands(case, meta["seeds"])
        m, n, k, pad, s = case["m"], case["n"], case["k"], case["pad"], case["stride"]
        alpha, beta = complex(*case["alpha"]), complex(*case["beta"])
        if case["kind"] < 2:
            alpha, beta = alpha.real, beta.real
        cbuf = np.array(c0.base, order="F", copy=True)
        ol.oracle_product64(case["kind"], case["opa"], case["opb"], m, n, k, alpha, a.base, a.shape[0] + pad, b.base,
                            b.shape[0] + pad, beta, cbuf, m + pad)
        assert np.array_equal(cbuf[m:, :], c0.base[m:, :])  # padding rows untouched
        got = cbuf[:m:s, :n:s]
        r = case_ratio(case, got, z[f"opstar_{idx}"], a, b, c0)
        if f"f77_{idx}" in z.files:
            r = max(r, case_ratio(case, got, z[f"f77_{idx}"], a, b, c0))
            c2 = np.array(c0.base, order="F", copy=True)
            ol.call_f77(ol.oracle_gemm(case["kind"]), case["opa"], case["opb"], m, n, k, alpha, a.base,
                        a.shape[0] + pad, b.base, b.shape[0] + pad, beta, c2, m + pad)
            assert np.array_equal(c2, cbuf)  # both oracle entry points are one algorithm
        worst = max(worst, r)
        assert r < RATIO_LIMIT, (idx, case, r)
    assert worst > 0.0


ref_scalar = ol.load_ref("novec")
ref_simd = ol.load_ref()


@pytest.mark.skipif(ref_scalar is None, reason="oracle/_ref not built (needs the reference tree at build time)")
def test_oracle_bitwise_vs_scalar_reference():
    """Eigen compiled with EIGEN_DONT_VECTORIZE and no FMA contraction uses the
    register tile, blocking and summation order the oracle restates: outputs must be
    identical bit for bit, for every kind and transpose form."""
    ref_scalar.ref_set_cache_sizes(32 * 1024, 256 * 1024, 2 * 1024 * 1024)
    rng = np.random.default_rng(7)
    for kind in range(4):
        dt = ol.DTYPES[kind]
        for (m, n, k) in [(5, 7, 9), (33, 65, 129), (100, 100, 100), (257, 131, 77), (64, 300, 1700), (301, 5, 1000)]:
            for ta in "NTC":
                for tb in "NTC":
                    ar, ac = (m, k) if ta == "N" else (k, m)
                    br, bc = (k, n) if tb == "N" else (n, k)

                    def rnd(r, c):
                        x = rng.uniform(-1, 1, (r + 1, c))
                        if kind >= 2:
                            x = x + 1j * rng.uniform(-1, 1, (r + 1, c))
                        return np.asfortranarray(x.astype(dt))
                    a, b, c0 = rnd(ar, ac), rnd(br, bc), rnd(m, n)
                    alpha, beta = (0.7 - 0.9j, 1.3 - 1.1j) if kind >= 2 else (0.7, 1.3)
                    c1, c2 = c0.copy(order="F"), c0.copy(order="F")
                    ol.call_f77(ol.oracle_gemm(kind), ta, tb, m, n, k, alpha, a, ar + 1, b, br + 1, beta, c1, m + 1)
                    ol.call_f77(getattr(ref_scalar, ol.SUFFIX[kind] + "gemm_"), ta, tb, m, n, k, alpha, a, ar + 1, b,
                                br + 1, beta, c2, m + 1)
                    assert np.array_equal(c1, c2), (kind, m, n, k, ta, tb)


@pytest.mark.skipif(ref_scalar is None, reason="oracle/_ref not built")
def test_blocking_sizes_match_reference():
    ref_scalar.ref_set_cache_sizes(32 * 1024, 256 * 1024, 2 * 1024 * 1024)
    for kind in range(4):
        for (k, m, n) in [(100, 100, 100), (2048, 2048, 2048), (500, 3000, 70), (4096, 17, 900),
                          (16384, 16384, 16384), (47, 47, 47), (320, 320, 320), (3, 5000, 5000), (1000, 8, 8)]:
            r = [ctypes.c_long(v) for v in (k, m, n)]
            o = [ctypes.c_long(v) for v in (k, m, n)]
            ref_scalar.ref_blocking_sizes(kind, *[ctypes.byref(x) for x in r])
            getattr(ol.oracle, "oracle_blocking_sizes_" + ol.SUFFIX[kind])(*[ctypes.byref(x) for x in o])
            assert [x.value for x in r] == [x.value for x in o], (kind, k, m, n)


@pytest.mark.skipif(ref_simd is None, reason="oracle/_ref not built")
@pytest.mark.parametrize("kind", range(4))
def test_restated_netlib_driver_passes_the_reference(kind):
    """The restated tester must accept the routine it was written for."""
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    fn = getattr(ref_simd, ol.SUFFIX[kind] + "gemm_")
    rep = ol.run_netlib_chk1(kind, fn, p)
    assert rep.ncalls == 17496 and not rep.fatal and rep.errmax < p["thresh"]
    if p["test_error_exits"]:
        assert ol.run_netlib_chke(kind, fn) == 0
