"""Parity of the device LU (eigen_amd/csrc/lu.hip, through the C ABI) with the LU
oracle: row transpositions (identical for float64; for float32 identical up to a
near-tie, see lu_checks.py for the rule), identical zero-pivot bookkeeping, factors
within a stated normwise tolerance, and the reconstruction residual the reference's
own test uses (test/lu.cpp:168: m1 ~ plu.reconstructedMatrix()).

Tolerance: max|LU_gpu - LU_oracle| <= 64 * eps * n * max|LU| (while the transpositions
agree) and ||P*A - L*U||_F <= 8 * eps * n * ||A||_F (eps = machine epsilon of the kind)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from eigen_amd import lu as ealu
from lu_checks import check_against_oracle
from test_lu_oracle import SHAPES, oracle_lu, residual

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_blocked_lu_vs_oracle(dt):
    eps = np.finfo(dt).eps
    for (m, n) in SHAPES + [(1024, 1024), (1500, 1100), (2100, 2100)]:
        a = np.zeros((m + 2, n), dtype=dt, order="F")
        a[:m] = eigen_amd.fill_uniform_host(dt, m, n, 5 + m)
        a[m:] = -1.0e10
        a1, a2 = a.copy(order="F"), a.copy(order="F")
        t1, nb1, r1 = oracle_lu(a1[:m])
        t2, nb2, r2 = ealu.blocked_lu(a2[:m])
        assert np.array_equal(a2[m:], a[m:]), "padding rows were written"
        assert r1 == r2 == -1
        first_diff = check_against_oracle(a[:m], a1[:m], t1, a2[:m], t2, exact_pivots=dt == np.float64)
        if first_diff is None:
            assert nb1 == nb2
        else:
            print(f"{dt.__name__} {m}x{n}: near-tie at column {first_diff}, transpositions differ from there")


def test_zero_pivot_and_getrf_conventions():
    rng = np.random.default_rng(3)
    a = np.asfortranarray(rng.uniform(-1, 1, (40, 40)))
    a[:, 7] = 0.0
    a1, a2, a3 = a.copy(order="F"), a.copy(order="F"), a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1)
    t2, nb2, r2 = ealu.blocked_lu(a2)
    assert r1 == r2 == 7 and np.array_equal(t1, t2) and nb1 == nb2
    ipiv, info = ealu.getrf(a3)
    assert info == 8 and np.array_equal(ipiv, t1 + 1)          # lapack/lu.cpp:34-38
    assert np.allclose(a2, a3)
    # the oracle's getrf agrees
    a4 = a.copy(order="F")
    ip4 = np.zeros(40, dtype=np.int32)
    inf4 = ctypes.c_int()
    ol.oracle.oracle_getrf_d(ctypes.byref(ctypes.c_int(40)), ctypes.byref(ctypes.c_int(40)), ctypes.c_void_p(a4.ctypes.data),
                             ctypes.byref(ctypes.c_int(40)), ctypes.c_void_p(ip4.ctypes.data), ctypes.byref(inf4))
    assert inf4.value == 8 and np.array_equal(ip4, ipiv)


def test_device_lu_8192_residual():
    """Device-resident factorisation at a size the CPU oracle does not finish in
    seconds: residual and structural properties only."""
    import torch
    from eigen_amd.device import DeviceMatrix
    n = 8192
    a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(77)
    a0 = a.as_strided().clone()
    transp, nb, fz = ealu.getrf_dev(a)
    assert fz == -1
    t = transp.cpu().numpy()
    assert (t >= np.arange(n)).all() and (t < n).all()          # partial pivoting picks a row at or below the diagonal
    lu = a.as_strided()
    l = torch.tril(lu, -1) + torch.eye(n, dtype=torch.float64, device="cuda")
    u = torch.triu(lu)
    assert float(l.abs().max()) <= 1.0                         # |l_ij| <= 1 is what partial pivoting guarantees
    perm = np.arange(n)
    for i, p in enumerate(t):
        if p != i:
            perm[i], perm[p] = perm[p], perm[i]
    pa = a0[torch.as_tensor(perm, device="cuda")]
    res = float(torch.linalg.norm(pa - l @ u) / torch.linalg.norm(a0))
    assert res <= 8 * np.finfo(np.float64).eps * n, res
