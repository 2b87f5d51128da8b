"""CPU tests of the symm / hemm / trmm oracles (oracle/trsm_oracle.c): the restated Netlib
?CHK2 and TRMM rows of ?CHK3 with the reference's parameter files, the golden results of
Eigen's own entry points, and the reference library itself when oracle/_ref is built.
Tolerance: Netlib test ratio < 16, i.e. |c - c_ref| <= 16 eps (|alpha| |A||B| + |beta||C0|)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")
RATIO_LIMIT = 16.0

ref_simd = ol.load_ref()
needs_ref = pytest.mark.skipif(ref_simd is None, reason="oracle/_ref not built (needs /root/reference at build time)")


def golden():
    z = np.load(os.path.join(HERE, "golden", "level3x_golden.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def abs1_of(kind):
    return (lambda z: np.abs(z.real) + np.abs(z.imag)) if kind >= 2 else np.abs


def run_case(lookup, c):
    """Runs one golden-style case through `lookup(name)` -> F77 function; returns
    (result, gauge) with gauge = |alpha| |A||B| + |beta||C0| over the full result."""
    kind, sfx = c["kind"], ol.SUFFIX[c["kind"]]
    cplx = kind >= 2
    alpha = complex(*c["alpha"]) if cplx else c["alpha"][0]
    beta = complex(*c["beta"]) if cplx else c["beta"][0]
    abs1 = abs1_of(kind)
    if c["op"] == "trmm":
        a, b = ol.trsm_operands(kind, c["side"], c["uplo"], c["diag"], c["m"], c["n"], c["seed"], c["pad"])
        b0 = np.array(b)
        ol.call_f77_trsm(lookup(sfx + "trmm_"), c["side"], c["uplo"], c["trans"], c["diag"], c["m"], c["n"], alpha, a,
                         a.shape[0] + c["pad"], b, c["m"] + c["pad"])
        t = abs1(ol.dense_op_triangle(np.asarray(a), c["uplo"], c["trans"], c["diag"])).astype(np.float64)
        g = abs1(np.asarray(alpha)) * (t @ abs1(b0) if c["side"] == "L" else abs1(b0) @ t)
        return np.array(b), np.where(g == 0, 1.0, g)
    herm = c["op"] == "hemm"
    a, b, cm = ol.symm_operands(kind, c["side"], c["uplo"], herm, c["m"], c["n"], c["seed"], c["pad"])
    c0 = np.array(cm)
    ol.call_f77_symm(lookup(sfx + c["op"] + "_"), c["side"], c["uplo"], c["m"], c["n"], alpha, a, a.shape[0] + c["pad"], b,
                     c["m"] + c["pad"], beta, cm, c["m"] + c["pad"])
    s = abs1(ol.dense_symmetric(np.asarray(a), c["uplo"], herm)).astype(np.float64)
    g = abs1(np.asarray(alpha)) * (s @ abs1(b) if c["side"] == "L" else abs1(b) @ s) + abs1(np.asarray(beta)) * abs1(c0)
    return np.array(cm), np.where(g == 0, 1.0, g)


def oracle_lookup(name):
    return getattr(ol.oracle, "oracle_" + name[1:-1] + "_" + name[0])


def check_against_golden(lookup):
    z, meta = golden()
    assert len(meta["cases"]) > 500
    for idx, c in enumerate(meta["cases"]):
        got, g = run_case(lookup, c)
        s = c["stride"]
        d = abs1_of(c["kind"])(got[::s, ::s].astype(np.complex128) - z[f"r{idx}"].astype(np.complex128))
        assert (d / (ol.EPS[c["kind"]] * g[::s, ::s])).max() < RATIO_LIMIT, (idx, c)


@pytest.mark.parametrize("kind", range(4))
def test_netlib_symm_hemm_trmm_known_answers_oracle(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    rep = ol.run_netlib_chk2(kind, oracle_lookup(ol.SUFFIX[kind] + "symm_"), False, p)
    assert rep.ncalls == 1296 and not rep.fatal and rep.errmax < p["thresh"] == RATIO_LIMIT
    if kind >= 2:
        rep = ol.run_netlib_chk2(kind, oracle_lookup(ol.SUFFIX[kind] + "hemm_"), True, p)
        assert rep.ncalls == 1296 and not rep.fatal and rep.errmax < p["thresh"]
    rep = ol.run_netlib_chk3_trmm(kind, oracle_lookup(ol.SUFFIX[kind] + "trmm_"), p)
    assert rep.ncalls == 2592 and not rep.fatal and rep.errmax < p["thresh"]


def test_oracle_vs_golden():
    check_against_golden(oracle_lookup)


@needs_ref
@pytest.mark.parametrize("kind", range(4))
def test_restated_drivers_pass_the_reference(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    sfx = ol.SUFFIX[kind]
    rep = ol.run_netlib_chk2(kind, getattr(ref_simd, sfx + "symm_"), False, p)
    assert rep.ncalls == 1296 and not rep.fatal and rep.errmax < p["thresh"]
    if kind >= 2:
        rep = ol.run_netlib_chk2(kind, getattr(ref_simd, sfx + "hemm_"), True, p)
        assert rep.ncalls == 1296 and not rep.fatal and rep.errmax < p["thresh"]
    rep = ol.run_netlib_chk3_trmm(kind, getattr(ref_simd, sfx + "trmm_"), p)
    assert rep.ncalls == 2592 and not rep.fatal and rep.errmax < p["thresh"]
