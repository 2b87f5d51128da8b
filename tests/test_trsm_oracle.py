"""CPU tests of the triangular-solve oracle (oracle/trsm_oracle.c): the restated Netlib
?CHK3 / ?CHKE rows of TRSM with the reference's own parameter files, the golden solutions
of Eigen's ?trsm_, and the reference library itself when oracle/_ref is built.

Tolerance: the Netlib test ratio |op(A) X - alpha B| / (eps * gauge) < 16 (THRESHOLD line of
?blat3.dat); oracle against reference solutions: max|x - x_ref| <= 64 * eps * na * max|x_ref|
on matrices whose triangle is well conditioned."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")
RATIO_LIMIT = 16.0
TRSM_CALLS = 2592  # what the Fortran ?CHK3 reports for ?TRSM with these parameter files

ref_simd = ol.load_ref()
needs_ref = pytest.mark.skipif(ref_simd is None, reason="oracle/_ref not built (needs /root/reference at build time)")


def golden():
    z = np.load(os.path.join(HERE, "golden", "trsm_golden.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def solve_with(fn, c):
    a, b = ol.trsm_operands(c["kind"], c["side"], c["uplo"], c["diag"], c["m"], c["n"], c["seed"], c["pad"])
    b0 = np.array(b)
    alpha = complex(*c["alpha"]) if c["kind"] >= 2 else c["alpha"][0]
    na = c["m"] if c["side"] == "L" else c["n"]
    ol.call_f77_trsm(fn, c["side"], c["uplo"], c["trans"], c["diag"], c["m"], c["n"], alpha, a, na + c["pad"], b,
                     c["m"] + c["pad"])
    return a, b0, np.array(b), alpha


@pytest.mark.parametrize("kind", range(4))
def test_netlib_trsm_known_answers_oracle(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    rep = ol.run_netlib_chk3_trsm(kind, ol.oracle_trsm(kind), p)
    assert rep.ncalls == TRSM_CALLS
    assert not rep.fatal and rep.nfail_same == 0 and rep.nfail_ratio == 0 and rep.nfail_xerbla == 0
    assert rep.errmax < p["thresh"] == RATIO_LIMIT


@pytest.mark.parametrize("kind", range(4))
def test_netlib_trsm_error_exits_oracle(kind):
    assert ol.run_netlib_chke_trsm(kind, ol.oracle_trsm(kind)) == 0


def test_oracle_vs_golden():
    z, meta = golden()
    assert len(meta["cases"]) > 200
    for idx, c in enumerate(meta["cases"]):
        a, b0, x, alpha = solve_with(ol.oracle_trsm(c["kind"]), c)
        want = z[f"x{idx}"]
        got = x[::c["stride"], ::c["stride"]]
        na = c["m"] if c["side"] == "L" else c["n"]
        assert np.abs(got - want).max() <= 64 * ol.EPS[c["kind"]] * max(na, 1) * max(np.abs(want).max(), 1e-30), (idx, c)
        assert ol.trsm_residual_ratio(c["kind"], c["side"], c["uplo"], c["trans"], c["diag"], alpha, a, b0, x) < RATIO_LIMIT


@needs_ref
@pytest.mark.parametrize("kind", range(4))
def test_restated_trsm_driver_passes_the_reference(kind):
    """The restated tester must accept the routine it was written for."""
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    fn = getattr(ref_simd, ol.SUFFIX[kind] + "trsm_")
    rep = ol.run_netlib_chk3_trsm(kind, fn, p)
    assert rep.ncalls == TRSM_CALLS and not rep.fatal and rep.errmax < p["thresh"]
    if p["test_error_exits"]:
        assert ol.run_netlib_chke_trsm(kind, fn) == 0


@needs_ref
def test_oracle_vs_reference_fresh_inputs():
    rng = np.random.default_rng(5)
    for kind in range(4):
        for _ in range(12):
            c = dict(kind=kind, side="LR"[rng.integers(2)], uplo="UL"[rng.integers(2)], trans="NTC"[rng.integers(3)],
                     diag="NU"[rng.integers(2)], m=int(rng.integers(1, 90)), n=int(rng.integers(1, 90)),
                     alpha=[0.7, -0.9 if kind >= 2 else 0.0], pad=int(rng.integers(3)), seed=int(rng.integers(1 << 30)))
            _, _, x_o, _ = solve_with(ol.oracle_trsm(kind), c)
            _, _, x_r, _ = solve_with(getattr(ref_simd, ol.SUFFIX[kind] + "trsm_"), c)
            na = c["m"] if c["side"] == "L" else c["n"]
            assert np.abs(x_o - x_r).max() <= 64 * ol.EPS[kind] * na * np.abs(x_r).max(), c
