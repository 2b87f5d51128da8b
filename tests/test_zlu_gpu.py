"""Parity of the complex device LU (eigen_amd/csrc/zlu.hip, through the C ABI) with the LU oracle,
which is pinned against the reference's partial_lu_impl<std::complex<..>>::blocked_lu in
test_lu_oracle.py.  Same rule and tolerances as for the real kinds (lu_checks.py): identical
transpositions for complex128, identical up to a near-tie for complex64,
max|LU_gpu - LU_oracle| <= 64 * eps * n * max|LU|, ||P*A - L*U||_F <= 8 * eps * n * ||A||_F."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from eigen_amd import lu as ealu
from lu_checks import check_against_oracle, check_structure, residual
from test_lu_oracle import SHAPES, oracle_lu, random_matrix

pytestmark = pytest.mark.gpu

KINDS = [np.complex128, np.complex64]


@pytest.mark.parametrize("dt", KINDS)
def test_blocked_lu_vs_oracle(dt):
    for (m, n) in SHAPES + [(1024, 1024), (1500, 1100), (2100, 2100)]:
        rng = np.random.default_rng(31 * m + n)
        a = np.zeros((m + 2, n), dtype=dt, order="F")
        a[:m] = random_matrix(rng, (m, n), dt)
        a[m:] = -1.0e10
        a1, a2 = a.copy(order="F"), a.copy(order="F")
        t1, nb1, r1 = oracle_lu(a1[:m])
        t2, nb2, r2 = ealu.blocked_lu(a2[:m])
        assert np.array_equal(a2[m:], a[m:]), "padding rows were written"
        assert r1 == r2 == -1
        first_diff = check_against_oracle(a[:m], a1[:m], t1, a2[:m], t2, exact_pivots=dt == np.complex128)
        if first_diff is None:
            assert nb1 == nb2
        else:
            print(f"{dt.__name__} {m}x{n}: near-tie at column {first_diff}, transpositions differ from there")


@pytest.mark.parametrize("dt", KINDS)
def test_zero_pivot_and_getrf_conventions(dt):
    rng = np.random.default_rng(3)
    a = random_matrix(rng, (40, 40), dt)
    a[:, 7] = 0.0
    a1, a2, a3, a4 = (a.copy(order="F") for _ in range(4))
    t1, nb1, r1 = oracle_lu(a1)
    t2, nb2, r2 = ealu.blocked_lu(a2)
    assert r1 == r2 == 7 and np.array_equal(t1, t2) and nb1 == nb2
    ipiv, info = ealu.getrf(a3)                                  # cgetrf_ / zgetrf_
    assert info == 8 and np.array_equal(ipiv, t1 + 1)            # lapack/lu.cpp:34-38
    assert np.array_equal(a2, a3)
    ip4 = np.zeros(40, dtype=np.int32)
    inf4 = ctypes.c_int()
    fn = ol.oracle.oracle_getrf_z if dt == np.complex128 else ol.oracle.oracle_getrf_c
    fn(ctypes.byref(ctypes.c_int(40)), ctypes.byref(ctypes.c_int(40)), ctypes.c_void_p(a4.ctypes.data),
       ctypes.byref(ctypes.c_int(40)), ctypes.c_void_p(ip4.ctypes.data), ctypes.byref(inf4))
    assert inf4.value == 8 and np.array_equal(ip4, ipiv)
    eps = np.finfo(dt).eps
    assert np.abs(a3 - a4).max() <= 64 * eps * 40 * np.abs(a4).max()


@pytest.mark.parametrize("dt", KINDS)
def test_row_major(dt):
    rng = np.random.default_rng(8)
    for (m, n) in [(33, 33), (300, 200), (200, 300), (700, 700)]:
        a = random_matrix(rng, (m, n), dt)
        ac = np.ascontiguousarray(a)
        af = a.copy(order="F")
        tc, nbc, rc = ealu.blocked_lu_rowmajor(ac)
        tf, nbf, rf = ealu.blocked_lu(af)
        assert np.array_equal(tc, tf) and nbc == nbf and rc == rf
        assert np.array_equal(ac, af)


@pytest.mark.parametrize("dt", KINDS)
def test_device_resident_4096(dt):
    """A size the oracle does not finish in seconds: structure and residual only."""
    from eigen_amd.device import DeviceMatrix
    n = 4096
    kind = eigen_amd.C64 if dt == np.complex128 else eigen_amd.C32
    a = DeviceMatrix(n, n, kind).fill_uniform(19)
    a0 = a.to_numpy()
    transp, nb, fz = ealu.getrf_dev(a)
    assert fz == -1
    lu = a.to_numpy()
    t = transp.cpu().numpy()
    check_structure(lu, t, n, n)
    assert residual(a0, lu, t) <= 8 * np.finfo(dt).eps * n
