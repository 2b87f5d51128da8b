"""CPU tests of the rank-2k oracles (oracle_syr2k_* / oracle_her2k_* in oracle/trsm_oracle.c): the
restated Netlib ?CHK5 / ?CHKE rows of SYR2K and HER2K with the reference's parameter files, golden
results of Eigen's ?syr2k_ / ?her2k_ (return values included), and the reference library itself
when oracle/_ref is built.  Tolerance: Netlib test ratio < 16; against reference results
|c - c_ref| <= 16 eps (2 |alpha| |A||B|^T-type gauge + |beta||C0|) componentwise."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")
RATIO_LIMIT = 16.0
R2K_CALLS = [1944, 1944, 1296, 1296]  # what ?CHK5 reports with these parameter files

ref_simd = ol.load_ref()
needs_ref = pytest.mark.skipif(ref_simd is None, reason="oracle/_ref not built (needs /root/reference at build time)")
FORMS = [(0, False), (1, False), (2, False), (3, False), (2, True), (3, True)]


def golden():
    z = np.load(os.path.join(HERE, "golden", "r2k_golden.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def scalars(c):
    alpha, beta = complex(*c["alpha"]), complex(*c["beta"])
    if c["kind"] < 2:
        return alpha.real, beta.real
    return alpha, (beta.real if c["herm"] else beta)


def run_case(fn, c):
    a, b, cm = ol.r2k_operands(c["kind"], c["herm"], c["uplo"], c["trans"], c["n"], c["k"], c["seed"], c["pad"])
    c0 = np.array(cm)
    alpha, beta = scalars(c)
    ld = max(a.shape[0], 1) + c["pad"]
    ret = ol.call_f77_r2k(fn, c["herm"], c["uplo"], c["trans"], c["n"], c["k"], alpha, a, ld, b, ld, beta, cm, c["n"] + c["pad"])
    return a, b, c0, np.array(cm), ret


def r2k_ratio(c, a, b, c0, got, want, stride=1):
    kind, n, herm = c["kind"], c["n"], c["herm"]
    abs1 = (lambda z: np.abs(z.real) + np.abs(z.imag)) if kind >= 2 else np.abs
    alpha, beta = scalars(c)
    s = stride
    i, j = np.indices((n, n))
    tri = ((i <= j) if c["uplo"] == "U" else (i >= j))[::s, ::s]
    gs = got[::s, ::s]
    assert np.array_equal(gs[~tri], c0[::s, ::s][~tri]), c
    if herm and (c["k"] == 0 or alpha == 0) and beta == 1.0:
        assert np.array_equal(gs, c0[::s, ::s]), c
        return 0.0
    c0r = np.array(c0)
    if herm:
        if not ((c["k"] == 0 or alpha == 0) and beta == 1.0):
            assert np.all(np.abs(np.diagonal(got).imag) <= 64 * ol.EPS[kind] * (1 + np.abs(np.diagonal(got).real))), c
        c0r[np.arange(n), np.arange(n)] = np.diagonal(c0).real
    aa, bb = abs1(np.asarray(a)).astype(np.float64), abs1(np.asarray(b)).astype(np.float64)
    prod = (aa @ bb.T + bb @ aa.T) if c["trans"] == "N" else (aa.T @ bb + bb.T @ aa)
    g = abs1(np.asarray(alpha)) * prod + abs1(np.asarray(beta)) * abs1(c0r)
    g = np.where(g == 0, 1.0, g)[::s, ::s]
    d = abs1(gs.astype(np.complex128) - want.astype(np.complex128)) / (ol.EPS[kind] * g)
    return float(d[tri].max()) if n else 0.0


@pytest.mark.parametrize("kind,herm", FORMS)
def test_netlib_r2k_known_answers_oracle(kind, herm):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    rep = ol.run_netlib_chk5(kind, ol.oracle_r2k(kind, herm), herm, p)
    assert rep.ncalls == R2K_CALLS[kind]
    assert not rep.fatal and rep.nfail_same == 0 and rep.nfail_ratio == 0 and rep.nfail_xerbla == 0
    assert rep.errmax < p["thresh"] == RATIO_LIMIT
    assert ol.run_netlib_chke_r2k(kind, ol.oracle_r2k(kind, herm), herm) == 0


def test_oracle_vs_golden():
    z, meta = golden()
    assert len(meta["cases"]) == 252
    for idx, c in enumerate(meta["cases"]):
        a, b, c0, got, ret = run_case(ol.oracle_r2k(c["kind"], c["herm"]), c)
        assert ret == c["ret"], (idx, c)
        assert r2k_ratio(c, a, b, c0, got, z[f"c{idx}"], c["stride"]) < RATIO_LIMIT, (idx, c)


@needs_ref
@pytest.mark.parametrize("kind,herm", FORMS)
def test_restated_r2k_driver_passes_the_reference(kind, herm):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    fn = getattr(ref_simd, ol.SUFFIX[kind] + ("her2k_" if herm else "syr2k_"))
    rep = ol.run_netlib_chk5(kind, fn, herm, p)
    assert rep.ncalls == R2K_CALLS[kind] and not rep.fatal and rep.errmax < p["thresh"]
    assert ol.run_netlib_chke_r2k(kind, fn, herm) == 0


@needs_ref
def test_oracle_vs_reference_fresh_inputs():
    rng = np.random.default_rng(29)
    for kind, herm in FORMS:
        for _ in range(8):
            trans = ("NC" if herm else ("NT" if kind >= 2 else "NTC"))
            c = dict(kind=kind, herm=herm, uplo="UL"[rng.integers(2)], trans=trans[rng.integers(len(trans))],
                     n=int(rng.integers(1, 80)), k=int(rng.integers(0, 90)), alpha=[0.7, -0.9 if kind >= 2 else 0.0],
                     beta=[float(rng.choice([1.3, 1.0, 0.0])), 0.0 if (herm or kind < 2) else -1.1], pad=int(rng.integers(3)),
                     seed=int(rng.integers(1 << 30)))
            a, b, c0, got, ret = run_case(ol.oracle_r2k(kind, herm), c)
            _, _, _, want, ret_ref = run_case(getattr(ref_simd, ol.SUFFIX[kind] + ("her2k_" if herm else "syr2k_")), c)
            assert ret == ret_ref, c
            assert r2k_ratio(c, a, b, c0, got, want) < RATIO_LIMIT, c
