"""Shared checks of the LU parity tests.

Pivot rule.  The oracle and the device run the same algorithm (first row of largest |a|, the modulus
for a complex scalar, Eigen/src/LU/PartialPivLU.h:375-377) but not the same roundings: the rank-16 and rank-256
updates are fused multiply-adds summed in another order.  Two candidates of a column whose
magnitudes differ by less than the rounding the column has accumulated may therefore swap
places.  What is demanded:
  * float64, complex128: identical transpositions (holds for every case on file);
  * float32, complex64: identical transpositions up to the first column j where they differ; there the
    device's choice must be a near-tie of the oracle's:
        |pivot_oracle| - |candidate_device| <= 8 * eps * (j + 1) * |pivot_oracle|
    (both magnitudes read from the oracle's factors: |l_rj| * |u_jj|, the row located through the
    oracle's later transpositions, which also move the finished part of L); after that column the
    two factorisations are different but equally valid ones, and only the properties every
    partial-pivoting LU has are checked: rows in range and at or below the diagonal,
    |l_ij| <= 1, the residual ||P A - L U||_F <= 8 * eps * n * ||A||_F (test/lu.cpp:168).
"""
import numpy as np


def perm_of(transp, m):
    perm = np.arange(m)
    for i, p in enumerate(transp):
        if p != i:
            perm[i], perm[p] = perm[p], perm[i]
    return perm


def residual(a0, lu, transp):
    """|| P*A - L*U ||_F / || A ||_F on the leading min(m,n) columns the blocked algorithm covers."""
    m, n = a0.shape
    size = min(m, n)
    wide = np.complex128 if np.iscomplexobj(a0) else np.float64
    pa = a0.astype(wide)[perm_of(transp, m)]
    l = np.tril(lu[:, :size].astype(wide), -1) + np.eye(m, size)
    u = np.triu(lu[:size, :].astype(wide))
    cols = n if m <= 16 else size          # the reference's blocked branch never touches columns >= size
    return np.linalg.norm((pa - l @ u)[:, :cols]) / np.linalg.norm(a0)


def check_structure(lu, transp, m, n):
    size = min(m, n)
    t = np.asarray(transp)
    assert (t >= np.arange(size)).all() and (t < m).all(), "a transposition names a row above the diagonal or out of range"
    slack = 4 * np.finfo(lu.dtype).eps if np.iscomplexobj(lu) else 0.0   # a complex quotient of modulus 1 is rounded
    assert np.abs(np.tril(lu[:, :size], -1)).max(initial=0.0) <= 1.0 + slack, "|l_ij| > 1"


def check_against_oracle(a0, lu_or, t_or, lu_dev, t_dev, exact_pivots):
    """a0: the matrix; (lu_or, t_or): the oracle's factors and transpositions; (lu_dev, t_dev): the
    device's.  Returns the first column where the transpositions differ (None: identical)."""
    m, n = a0.shape
    dt = a0.dtype
    eps = np.finfo(dt).eps
    size = min(m, n)
    check_structure(lu_dev, t_dev, m, n)
    assert residual(a0, lu_dev, t_dev) <= 8 * eps * max(m, n), (m, n)
    diff = np.nonzero(np.asarray(t_or) != np.asarray(t_dev))[0]
    if diff.size == 0:
        scale = np.abs(lu_or).max()
        err = np.abs(lu_or.astype(np.complex128) - lu_dev.astype(np.complex128)).max()
        assert err <= 64 * eps * max(m, n) * scale, (m, n, err)
        return None
    j = int(diff[0])
    assert not exact_pivots, f"transpositions differ from column {j} on ({m}x{n} {dt.__name__})"
    # the device's row, located in the oracle's factors: rows j and t_or[j] have been exchanged
    # there, and every later transposition has moved whole rows, the finished columns included
    p_or, p_dev = int(t_or[j]), int(t_dev[j])
    row = p_or if p_dev == j else p_dev
    for i in range(j + 1, size):
        p = int(t_or[i])
        if row == i:
            row = p
        elif row == p:
            row = i
    piv = float(abs(lu_or[j, j]))
    cand = float(abs(lu_or[row, j])) * piv
    assert piv - cand <= 8 * eps * (j + 1) * piv, \
        f"column {j}: device pivot row {p_dev} (|a| = {cand}) is no near-tie of the oracle's row {p_or} (|a| = {piv})"
    assert size > 0
    return j
