"""CPU tests of the symmetric rank-k oracle (oracle_syrk_* in oracle/trsm_oracle.c): the
restated Netlib ?CHK4 / ?CHKE rows of SYRK with the reference's parameter files, the golden
results of Eigen's ?syrk_, and the reference library itself when oracle/_ref is built.
Tolerance: Netlib test ratio < 16; against reference results |c - c_ref| <= 16 eps (|alpha|
(|A||A|^T) + |beta||C0|) componentwise."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")
RATIO_LIMIT = 16.0
SYRK_CALLS = [1944, 1944, 1296, 1296]  # what ?CHK4 reports for ?SYRK with these parameter files

ref_simd = ol.load_ref()
needs_ref = pytest.mark.skipif(ref_simd is None, reason="oracle/_ref not built (needs /root/reference at build time)")


def golden():
    z = np.load(os.path.join(HERE, "golden", "syrk_golden.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def scalars(c):
    if c["kind"] >= 2:
        return complex(*c["alpha"]), complex(*c["beta"])
    return c["alpha"][0], c["beta"][0]


def run_case(fn, c):
    a, cm = ol.syrk_operands(c["kind"], c["uplo"], c["trans"], c["n"], c["k"], c["seed"], c["pad"])
    c0 = np.array(cm)
    alpha, beta = scalars(c)
    ol.call_f77_syrk(fn, c["uplo"], c["trans"], c["n"], c["k"], alpha, a, a.shape[0] + c["pad"], beta, cm, c["n"] + c["pad"])
    return a, c0, np.array(cm), alpha, beta


def syrk_ratio(kind, uplo, trans, alpha, beta, a, c0, got, want):
    """Componentwise |got - want| / (eps * gauge) on the triangle; outside it `got` must equal c0."""
    abs1 = (lambda z: np.abs(z.real) + np.abs(z.imag)) if kind >= 2 else np.abs
    aa = abs1(np.asarray(a)).astype(np.float64)
    g = abs1(np.asarray(alpha)) * (aa @ aa.T if trans == "N" else aa.T @ aa) + abs1(np.asarray(beta)) * abs1(c0)
    g = np.where(g == 0, 1.0, g)
    n = c0.shape[0]
    i, j = np.indices((n, n))
    tri = (i <= j) if uplo == "U" else (i >= j)
    assert np.array_equal(got[~tri], c0[~tri])
    d = abs1(got.astype(np.complex128) - want.astype(np.complex128)) / (ol.EPS[kind] * g)
    return float(d[tri].max()) if n else 0.0


@pytest.mark.parametrize("kind", range(4))
def test_netlib_syrk_known_answers_oracle(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    rep = ol.run_netlib_chk4_syrk(kind, ol.oracle_syrk(kind), p)
    assert rep.ncalls == SYRK_CALLS[kind]
    assert not rep.fatal and rep.nfail_same == 0 and rep.nfail_ratio == 0 and rep.nfail_xerbla == 0
    assert rep.errmax < p["thresh"] == RATIO_LIMIT
    assert ol.run_netlib_chke_syrk(kind, ol.oracle_syrk(kind)) == 0


def test_oracle_vs_golden():
    z, meta = golden()
    for idx, c in enumerate(meta["cases"]):
        a, c0, got, alpha, beta = run_case(ol.oracle_syrk(c["kind"]), c)
        s = c["stride"]
        want_s = z[f"c{idx}"]
        # strided comparison against the stored sample
        abs1 = (lambda v: np.abs(v.real) + np.abs(v.imag)) if c["kind"] >= 2 else np.abs
        aa = abs1(np.asarray(a)).astype(np.float64)
        g = abs1(np.asarray(alpha)) * (aa @ aa.T if c["trans"] == "N" else aa.T @ aa) + abs1(np.asarray(beta)) * abs1(c0)
        g = np.where(g == 0, 1.0, g)[::s, ::s]
        n = c["n"]
        i, j = np.indices((n, n))
        tri = ((i <= j) if c["uplo"] == "U" else (i >= j))[::s, ::s]
        gs = got[::s, ::s]
        assert np.array_equal(gs[~tri], want_s[~tri]), (idx, c)
        d = abs1(gs.astype(np.complex128) - want_s.astype(np.complex128)) / (ol.EPS[c["kind"]] * g)
        assert d[tri].max() < RATIO_LIMIT, (idx, c)


@needs_ref
@pytest.mark.parametrize("kind", range(4))
def test_restated_syrk_driver_passes_the_reference(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    fn = getattr(ref_simd, ol.SUFFIX[kind] + "syrk_")
    rep = ol.run_netlib_chk4_syrk(kind, fn, p)
    assert rep.ncalls == SYRK_CALLS[kind] and not rep.fatal and rep.errmax < p["thresh"]
    if p["test_error_exits"]:
        assert ol.run_netlib_chke_syrk(kind, fn) == 0


@needs_ref
def test_oracle_vs_reference_fresh_inputs():
    rng = np.random.default_rng(9)
    for kind in range(4):
        for _ in range(10):
            c = dict(kind=kind, uplo="UL"[rng.integers(2)], trans=("NT" if kind >= 2 else "NTC")[rng.integers(2 if kind >= 2 else 3)],
                     n=int(rng.integers(1, 80)), k=int(rng.integers(0, 90)), alpha=[0.7, -0.9 if kind >= 2 else 0.0],
                     beta=[1.3, -1.1 if kind >= 2 else 0.0], pad=int(rng.integers(3)), seed=int(rng.integers(1 << 30)))
            a, c0, got, alpha, beta = run_case(ol.oracle_syrk(kind), c)
            _, _, want, _, _ = run_case(getattr(ref_simd, ol.SUFFIX[kind] + "syrk_"), c)
            assert syrk_ratio(kind, c["uplo"], c["trans"], alpha, beta, a, c0, got, want) < RATIO_LIMIT, c
