"""CPU tests of the Hermitian rank-k oracle (oracle_herk_* in oracle/trsm_oracle.c): the restated
HERK rows of the Netlib ZCHK4 / ZCHKE with the reference's parameter files, golden results of
Eigen's ?herk_, and the reference library itself when oracle/_ref is built.
Tolerance: Netlib test ratio < 16; against reference results |c - c_ref| <= 16 eps (|alpha|
(|A||A|^H) + |beta||C0|) componentwise (the rogue imaginary part of C0's diagonal left out)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")
RATIO_LIMIT = 16.0
HERK_CALLS = 1296  # 6 x 6 sizes, 2 trans, 2 uplo, 3 x 3 scalars

ref_simd = ol.load_ref()
needs_ref = pytest.mark.skipif(ref_simd is None, reason="oracle/_ref not built (needs /root/reference at build time)")


def golden():
    z = np.load(os.path.join(HERE, "golden", "herk_golden.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def run_case(fn, c):
    a, cm = ol.herk_operands(c["kind"], c["uplo"], c["trans"], c["n"], c["k"], c["seed"], c["pad"])
    c0 = np.array(cm)
    ol.call_f77_herk(fn, c["uplo"], c["trans"], c["n"], c["k"], c["alpha"], a, max(a.shape[0], 1) + c["pad"], c["beta"], cm,
                     c["n"] + c["pad"])
    return a, c0, np.array(cm)


def herk_ratio(c, a, c0, got, want, stride=1):
    """Componentwise |got - want| / (eps * gauge) on the triangle; outside it `got` must equal c0.
    When the call has nothing to do (no product and beta == 1) every bit must be c0's."""
    kind, n = c["kind"], c["n"]
    abs1 = lambda z: np.abs(z.real) + np.abs(z.imag)
    s = stride
    i, j = np.indices((n, n))
    tri = ((i <= j) if c["uplo"] == "U" else (i >= j))[::s, ::s]
    gs = got[::s, ::s]
    assert np.array_equal(gs[~tri], c0[::s, ::s][~tri]), c
    if (c["k"] == 0 or c["alpha"] == 0.0) and c["beta"] == 1.0:
        assert np.array_equal(gs, c0[::s, ::s]), c
        return 0.0
    assert np.all(np.diagonal(got).imag == 0), c
    aa = abs1(np.asarray(a)).astype(np.float64)
    c0r = np.array(c0)
    c0r[np.arange(n), np.arange(n)] = np.diagonal(c0).real   # the diagonal's imaginary part is not an input
    g = abs(c["alpha"]) * (aa @ aa.T if c["trans"] == "N" else aa.T @ aa) + abs(c["beta"]) * abs1(c0r)
    g = np.where(g == 0, 1.0, g)[::s, ::s]
    d = abs1(gs.astype(np.complex128) - want.astype(np.complex128)) / (ol.EPS[kind] * g)
    return float(d[tri].max()) if n else 0.0


@pytest.mark.parametrize("kind", [2, 3])
def test_netlib_herk_known_answers_oracle(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    rep = ol.run_netlib_chk4_herk(kind, ol.oracle_herk(kind), p)
    assert rep.ncalls == HERK_CALLS
    assert not rep.fatal and rep.nfail_same == 0 and rep.nfail_ratio == 0 and rep.nfail_xerbla == 0
    assert rep.errmax < p["thresh"] == RATIO_LIMIT
    assert ol.run_netlib_chke_herk(kind, ol.oracle_herk(kind)) == 0


def test_oracle_vs_golden():
    z, meta = golden()
    assert len(meta["cases"]) == 80
    for idx, c in enumerate(meta["cases"]):
        a, c0, got = run_case(ol.oracle_herk(c["kind"]), c)
        assert herk_ratio(c, a, c0, got, z[f"c{idx}"], c["stride"]) < RATIO_LIMIT, (idx, c)


@needs_ref
@pytest.mark.parametrize("kind", [2, 3])
def test_restated_herk_driver_passes_the_reference(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    fn = getattr(ref_simd, ol.SUFFIX[kind] + "herk_")
    rep = ol.run_netlib_chk4_herk(kind, fn, p)
    assert rep.ncalls == HERK_CALLS and not rep.fatal and rep.errmax < p["thresh"]
    assert ol.run_netlib_chke_herk(kind, fn) == 0


@needs_ref
def test_oracle_vs_reference_fresh_inputs():
    rng = np.random.default_rng(19)
    for kind in (2, 3):
        for _ in range(12):
            c = dict(kind=kind, uplo="UL"[rng.integers(2)], trans="NC"[rng.integers(2)], n=int(rng.integers(1, 80)),
                     k=int(rng.integers(0, 90)), alpha=float(rng.choice([0.7, -1.0, 0.0])), beta=float(rng.choice([1.3, 1.0, 0.0])),
                     pad=int(rng.integers(3)), seed=int(rng.integers(1 << 30)))
            a, c0, got = run_case(ol.oracle_herk(kind), c)
            _, _, want = run_case(getattr(ref_simd, ol.SUFFIX[kind] + "herk_"), c)
            assert herk_ratio(c, a, c0, got, want) < RATIO_LIMIT, c
