"""Pins the ?getrs oracle (oracle_getrs_* in oracle/trsm_oracle_impl.inc) against the reference: the three
Eigen expressions of EIGEN_LAPACK_FUNC(getrs) (lapack/lu.cpp:62-84) evaluated by the reference compiled
under oracle/_ref (ref_getrs in oracle/ref_harness.cpp), on factors from the LU oracle.

Tolerance: both solve the same triangular systems with different summation orders, so the solutions
differ by the conditioning of the triangles: |x_oracle - x_ref| <= 16 * eps * n * cond(A) * max|x|, and each
satisfies ||op(A) X - B||_F <= 8 * eps * n * ||A||_F * ||X||_F (eps of the kind)."""
import numpy as np
import pytest

import oracle_lib as ol
from test_lu_oracle import oracle_lu, random_matrix

CASES = [(1, 1), (5, 3), (16, 16), (40, 7), (130, 33), (300, 64), (64, 300)]
OPS = {"N": 0, "T": 1, "C": 2}


def op_of(a, trans):
    return a if trans == "N" else (a.T if trans == "T" else a.conj().T)


def factor(dt, n, seed):
    rng = np.random.default_rng(seed)
    a = random_matrix(rng, (n, n), dt)
    lu = a.copy(order="F")
    t, _, fz = oracle_lu(lu)
    assert fz == -1
    return a, lu, t.astype(np.int32)


def backward_error(a, trans, x, b):
    wide = np.complex128 if np.iscomplexobj(a) else np.float64
    aw, xw, bw = a.astype(wide), x.astype(wide), b.astype(wide)
    return np.linalg.norm(op_of(aw, trans) @ xw - bw) / (np.linalg.norm(aw) * np.linalg.norm(xw))


@pytest.mark.parametrize("variant", ["novec", "v3"])
@pytest.mark.parametrize("kind", range(4))
def test_getrs_oracle_vs_reference(variant, kind):
    ref = ol.load_ref(variant)
    if ref is None or not hasattr(ref, "ref_getrs"):
        pytest.skip("oracle/_ref not built")
    dt = ol.DTYPES[kind]
    eps = np.finfo(dt).eps
    for (n, nrhs) in CASES:
        a, lu, t = factor(dt, n, 100 * n + nrhs)
        cond = np.linalg.cond(a.astype(np.complex128 if kind >= 2 else np.float64))
        for trans in "NTC":
            rng = np.random.default_rng(n + nrhs)
            b0 = random_matrix(rng, (n + 3, nrhs), dt)
            b0[n:] = -1.0e10
            b1, b2 = b0.copy(order="F"), b0.copy(order="F")
            ipiv = (t + 1).astype(np.int32)
            info = ol.call_f77_getrs(ol.oracle_getrs(kind), trans, n, nrhs, lu, n, ipiv, b1, n + 3)
            assert info == 0 and np.array_equal(ipiv, t + 1)            # ipiv restored (lu.cpp:85-86)
            assert np.array_equal(b1[n:], b0[n:])                       # rows beyond n untouched
            t0 = t.copy()
            ref.ref_getrs(kind, OPS[trans], n, nrhs, lu.ctypes.data, n, t0.ctypes.data, b2.ctypes.data, n + 3)
            x1, x2 = b1[:n], b2[:n]
            assert np.abs(x1 - x2).max() <= 16 * eps * n * cond * np.abs(x2).max(), (n, nrhs, trans)
            assert backward_error(a, trans, x1, b0[:n]) <= 8 * eps * n
            assert backward_error(a, trans, x2, b0[:n]) <= 8 * eps * n


@pytest.mark.parametrize("kind", range(4))
def test_getrs_oracle_argument_errors(kind):
    """info = -1 (trans), -2 (n), -3 (nrhs), -5 (lda), -8 (ldb) through xerbla_ (lapack/lu.cpp:48-58)."""
    dt = ol.DTYPES[kind]
    a = np.zeros((4, 4), dtype=dt, order="F")
    b = np.ones((4, 2), dtype=dt, order="F")
    ipiv = np.arange(1, 5, dtype=np.int32)
    name = "SDCZ"[kind] + "GETRS"
    for (trans, n, nrhs, lda, ldb, code) in [("X", 4, 2, 4, 4, 1), ("N", -1, 2, 4, 4, 2), ("N", 4, -1, 4, 4, 3),
                                             ("T", 4, 2, 3, 4, 5), ("C", 4, 2, 4, 3, 8)]:
        ol.reset_xerbla()
        info = ol.call_f77_getrs(ol.oracle_getrs(kind), trans, n, nrhs, a, lda, ipiv, b, ldb)
        assert info == -code and ol.last_xerbla() == (code, name)
        assert np.all(b == 1) and np.array_equal(ipiv, np.arange(1, 5))
    ol.reset_xerbla()
