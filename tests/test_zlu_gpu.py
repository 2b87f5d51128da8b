"""Parity of the device LU for the complex kinds (eigen_amd/csrc/lu.hip instantiated over a (re, im) scalar,
through the C ABI) with the LU oracle, which is pinned against the reference's
partial_lu_impl<std::complex<..>>::blocked_lu in test_lu_oracle.py.  Same rule and tolerances as for the real kinds (lu_checks.py): identical
transpositions for complex128, identical up to a near-tie for complex64,
max|LU_gpu - LU_oracle| <= 64 * eps * n * max|LU|, ||P*A - L*U||_F <= 8 * eps * n * ||A||_F."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from eigen_amd import lu as ealu
from lu_checks import check_against_oracle, check_structure, residual
from test_lu_config_gpu import env
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


# Every base case and every switch of the driver, for the complex scalars: the per-column kernel chain, the one-launch
# base case on short panels, the inverse built behind the panel, in-place solves of the next block's columns, the far
# part of a block's update as a launch of its own, row operations and rank update as two launches, substitution in
# place of every inverse, forced wide outer steps, the trailing update on two streams, the one-launch base case without the
# candidates' rows read beside the poll.
SWITCHES = [dict(EIGAMD_LU_FUSED=0, EIGAMD_LU_SINGLE=0), dict(EIGAMD_LU_SINGLE=0), dict(EIGAMD_LU_INVERSE_ON_THE_WAY=0),
            dict(EIGAMD_LU_SOLVE_TMP=0), dict(EIGAMD_LU_FAR_RIDES=0), dict(EIGAMD_LU_FUSED_ROWOPS=0),
            dict(EIGAMD_LU_INVERSE_LIMIT=0), dict(EIGAMD_LU_WIDE=512), dict(EIGAMD_LU_SPLIT_MIN=256, EIGAMD_LU_WIDE=256),
            dict(EIGAMD_LU_WIDE_POLL=0), dict(EIGAMD_LU_WIDE_POLL=0, EIGAMD_LU_SINGLE=0)]


_ORACLE = {}


def matrix_and_oracle(dt, m, n):
    """(matrix, oracle's factors, transpositions, exchanges, first zero pivot), worked out once per kind and shape."""
    if (dt, m, n) not in _ORACLE:
        a = random_matrix(np.random.default_rng(17 * m + n), (m, n), dt).copy(order="F")
        a1 = a.copy(order="F")
        _ORACLE[(dt, m, n)] = (a, a1) + tuple(oracle_lu(a1))
    return _ORACLE[(dt, m, n)]


@pytest.mark.parametrize("dt", KINDS)
@pytest.mark.parametrize("knobs", SWITCHES, ids=lambda k: ",".join(f"{a[10:]}={b}" for a, b in k.items()))
def test_driver_switches_vs_oracle(dt, knobs):
    for (m, n) in [(2304, 2304), (1500, 1100)]:
        a, a1, t1, nb1, r1 = matrix_and_oracle(dt, m, n)
        a2 = a.copy(order="F")
        with env(**knobs):
            t2, nb2, r2 = ealu.blocked_lu(a2)
        assert r1 == r2 == -1
        if check_against_oracle(a, a1, t1, a2, t2, exact_pivots=dt == np.complex128) is None:
            assert nb1 == nb2


@pytest.mark.parametrize("dt", KINDS)
@pytest.mark.parametrize("knobs", [dict(EIGAMD_LU_FUSED=0, EIGAMD_LU_SINGLE=0), dict(EIGAMD_LU_SINGLE=0), dict()])
def test_column_of_nans_takes_the_diagonal_entry(dt, knobs):
    n = 600
    a = random_matrix(np.random.default_rng(5), (n, n), dt).copy(order="F")
    a[:, 5] = np.nan
    with env(**knobs):
        t, nb, fz = ealu.blocked_lu(a)
    assert t[5] == 5 and (t >= np.arange(n)).all() and (t < n).all()


@pytest.mark.parametrize("dt", KINDS)
def test_entry_that_does_not_wait_and_the_shared_form(dt):
    """eigamd_getrf_dev_async and eigamd_getrf_dev_shared take the complex kinds as they take the real ones."""
    import torch
    from eigen_amd.device import DeviceMatrix
    m, n = 1300, 1300
    a = random_matrix(np.random.default_rng(11), (m, n), dt).copy(order="F")
    a1 = a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1)
    d = DeviceMatrix.from_numpy(a)
    transp = torch.zeros(n, dtype=torch.int32, device="cuda")
    info = torch.zeros(3, dtype=torch.int32, device="cuda")
    ealu.getrf_dev_async(d, transp, info)
    torch.cuda.synchronize()
    fz, nb, status = info.cpu().tolist()
    assert (fz, status) == (-1, 0)
    if check_against_oracle(a, a1, t1, d.to_numpy(), transp.cpu().numpy(), exact_pivots=dt == np.complex128) is None:
        assert nb == nb1
    d2 = DeviceMatrix.from_numpy(a)
    t2, nb2, fz2 = ealu.getrf_dev(d2, shared=True)
    assert fz2 == -1
    check_against_oracle(a, a1, t1, d2.to_numpy(), t2.cpu().numpy(), exact_pivots=dt == np.complex128)


def test_tall_panel_through_the_exchange_area():
    """Enough rows that a panel's base case is the one-launch kernel with many workgroups (complex128: 5000 rows = 20)."""
    m, n = 5000, 300
    a = random_matrix(np.random.default_rng(23), (m, n), np.complex128).copy(order="F")
    a1, a2 = a.copy(order="F"), a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1)
    t2, nb2, r2 = ealu.blocked_lu(a2)
    check_against_oracle(a, a1, t1, a2, t2, exact_pivots=True)
    assert nb1 == nb2 and r1 == r2 == -1


@pytest.mark.parametrize("dt,n", [(np.complex128, 20480), (np.complex64, 24576)])
def test_panels_taller_than_the_default_panel_units_hold(dt, n):
    """80 and 96 workgroups of the one-launch base case at two per compute unit: more than the 32 units the panel stream
    has by default, so it gets more -- in steps of 32, because the workgroups of a launch are dealt evenly over the shader
    engines and only such masks give every engine the same number of units (48 units never got the 96 workgroups resident
    together).  Structure of the result and the residual on sampled rows."""
    import torch
    from eigen_amd.device import DeviceMatrix
    from test_lu_config_gpu import sampled_lu_residual
    kind = eigen_amd.C64 if dt == np.complex128 else eigen_amd.C32
    a = DeviceMatrix(n, n, kind).fill_uniform(31)
    transp, nb, fz = ealu.getrf_dev(a)
    t = transp.cpu().numpy()
    assert fz == -1 and (t >= np.arange(n)).all() and (t < n).all()
    a0 = DeviceMatrix(n, n, kind).fill_uniform(31)
    assert sampled_lu_residual(n, a, a0, t, nblocks=3) <= 8 * np.finfo(dt).eps * n
    del a, a0
    torch.cuda.empty_cache()
