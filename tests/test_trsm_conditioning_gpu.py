"""?trsm_ / eigamd_trsm_dev on triangles that are NOT well conditioned.

The library solves a diagonal block by multiplying with its explicit inverse only while
max|inv(T)| * max|T| stays below a limit (8); other blocks are solved by substitution on the
device, the reference's own method (Eigen/src/Core/products/TriangularSolverMatrix.h:40-...,
which is backward stable row by row).  Checked here:

* the substitution kernel alone (limit forced to 0) against the oracle on every side / uplo /
  trans / diag form and kind: it is the reference's recurrence, so the solutions agree to
  rounding on well-conditioned operands;
* adversarial triangles with the shipped limit -- unit lower with -1 below the diagonal (inverse
  entries up to 2^(n-2)), graded diagonals (condition 1e8 .. 1e12 per block), dense random
  triangles (exponentially ill conditioned): the componentwise residual
      |op(T) X - alpha B| / (eps * (|op(T)| |X| + |alpha| |B|))
  of the device solution must stay within 4x the oracle's (and below the Netlib limit of 16
  when the oracle's is);
* what the guard buys: with the guard switched off the same operands give a residual orders of
  magnitude above the oracle's (asserted, so that the test fails if the guard stops mattering).
"""
import os

import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from eigen_amd.device import DeviceMatrix, trsm_dev
from test_lu_config_gpu import env

pytestmark = pytest.mark.gpu


def solve_both(kind, side, uplo, trans, diag, alpha, a, b):
    """(device solution, oracle solution) of one ?trsm_ problem on host arrays."""
    m, n = b.shape
    na = a.shape[0]
    want = np.array(b, order="F")
    ol.call_f77_trsm(ol.oracle_trsm(kind), side, uplo, trans, diag, m, n, alpha, a, na, want, m)
    da, db = DeviceMatrix.from_numpy(np.array(a)), DeviceMatrix.from_numpy(np.array(b))
    trsm_dev(side, uplo, trans, diag, alpha, da, db)
    return db.to_numpy(), want


@pytest.mark.parametrize("kind", range(4))
def test_substitution_kernel_vs_oracle_all_forms(kind):
    alpha = (0.7 - 0.9j) if kind >= 2 else 0.7
    seed = 140
    with env(EIGAMD_TRSM_GROWTH_LIMIT=0):
        for side in "LR":
            for uplo in "UL":
                for trans in "NTC":
                    for diag in "NU":
                        seed += 1
                        m, n = (300, 170) if side == "L" else (170, 300)
                        a, b = ol.trsm_operands(kind, side, uplo, diag, m, n, seed)
                        got, want = solve_both(kind, side, uplo, trans, diag, alpha, np.array(a), np.array(b))
                        na = a.shape[0]
                        assert np.abs(got - want).max() <= 64 * ol.EPS[kind] * na * np.abs(want).max(), (side, uplo, trans, diag)
                        assert ol.trsm_residual_ratio(kind, side, uplo, trans, diag, alpha, a, b, got) < 16.0


def adversarial(name, dt, na, rng):
    """Lower triangular na x na test matrices (diag 'N' unless stated by the caller)."""
    if name == "minus_ones":          # unit lower, -1 below the diagonal: inv(T)(i, j) = 2^(i-j-1)
        return np.asfortranarray(np.tril(-np.ones((na, na)), -1) + np.eye(na)).astype(dt)
    if name == "graded":              # rows scaled over 10 decades per 128 rows (5 in single precision)
        t = np.tril(rng.uniform(-1, 1, (na, na)), -1) * 0.05 + np.eye(na)
        decades = 5.0 if dt == np.float32 else 10.0
        d = 10.0 ** (-decades * (np.arange(na) % 128) / 127.0)
        return np.asfortranarray(d[:, None] * t).astype(dt)
    if name == "random":              # dense random triangle, diagonal away from zero
        t = np.tril(rng.uniform(-1, 1, (na, na)), -1)
        return np.asfortranarray(t + np.diag(np.where(rng.uniform(size=na) < 0.5, -1.0, 1.0) * rng.uniform(0.5, 1.0, na))).astype(dt)
    raise ValueError(name)


CASES = [("minus_ones", 64), ("minus_ones", 200), ("minus_ones", 512), ("graded", 384), ("graded", 700), ("random", 300),
         ("random", 600)]


@pytest.mark.parametrize("kind", [1, 0, 3])
@pytest.mark.parametrize("name,na", CASES)
def test_adversarial_triangles_residual_vs_oracle(kind, name, na):
    dt = ol.DTYPES[kind]
    rng = np.random.default_rng(na + kind)
    t = adversarial(name, np.float32 if kind == 0 else np.float64, na, rng).astype(dt)
    if kind >= 2:
        t = (t + 1j * np.tril(rng.uniform(-0.3, 0.3, (na, na)), -1)).astype(dt)
    nrhs = 130
    worst = 0.0
    for side, uplo, trans in [("L", "L", "N"), ("L", "U", "T"), ("R", "L", "N"), ("R", "U", "N")]:
        a = np.asfortranarray(t if uplo == "L" else t.T)      # the stored triangle
        m, n = (na, nrhs) if side == "L" else (nrhs, na)
        # a right-hand side that a small solution produces: op(T) X0 with X0 of order one, so that
        # the exact solution is small against |inv(T)| |B| wherever T is ill conditioned
        x0 = rng.uniform(-1, 1, (m, n)).astype(dt)
        top = ol.dense_op_triangle(a, uplo, trans, "N")
        b = np.asfortranarray((top @ x0) if side == "L" else (x0 @ top)).astype(dt)
        if not np.isfinite(b).all():
            pytest.skip("right-hand side overflows in this precision")
        got, want = solve_both(kind, side, uplo, trans, "N", 1.0, a, b)
        if not np.isfinite(want).all():
            print(f"{name} {na} kind {kind} {side}{uplo}{trans}: the oracle's solution overflows; skipped")
            continue
        r_dev = ol.trsm_residual_ratio(kind, side, uplo, trans, "N", 1.0, a, b, got)
        r_or = ol.trsm_residual_ratio(kind, side, uplo, trans, "N", 1.0, a, b, want)
        print(f"{name} {na} kind {kind} {side}{uplo}{trans}: residual ratio device {r_dev:.3g} oracle {r_or:.3g}")
        assert np.isfinite(got).all()
        assert r_dev <= max(16.0, 4.0 * r_or), (name, na, side, uplo, trans, r_dev, r_or)
        worst = max(worst, r_dev)
    assert worst < 64.0


def test_guard_off_shows_what_it_buys():
    """The same ill-conditioned operands with the guard switched off (every block through its
    explicit inverse): the residual leaves the oracle's by orders of magnitude."""
    kind, na = 1, 512
    rng = np.random.default_rng(1)
    a = adversarial("random", np.float64, na, rng)
    x0 = rng.uniform(-1, 1, (na, 100))
    b = np.asfortranarray(a @ x0)
    with env(EIGAMD_TRSM_GROWTH_LIMIT="inf"):
        got_inv, want = solve_both(kind, "L", "L", "N", "N", 1.0, a, b)
    got, _ = solve_both(kind, "L", "L", "N", "N", 1.0, a, b)
    r_inv = ol.trsm_residual_ratio(kind, "L", "L", "N", "N", 1.0, a, b, got_inv)
    r_dev = ol.trsm_residual_ratio(kind, "L", "L", "N", "N", 1.0, a, b, got)
    r_or = ol.trsm_residual_ratio(kind, "L", "L", "N", "N", 1.0, a, b, want)
    print(f"residual ratio: explicit inverses {r_inv:.3g}, guarded {r_dev:.3g}, oracle {r_or:.3g}")
    assert r_dev <= max(16.0, 4.0 * r_or)
    assert r_inv > 100.0 * max(r_dev, 1.0)
