"""Parity of ?getrs_ / eigamd_getrs_dev (row exchanges + the library's two triangular solves on the
device) with the getrs oracle, on factors from the device LU and from the LU oracle.

Tolerance (floating point): both sides solve the same two triangular systems in different summation
orders: |x_gpu - x_oracle| <= 16 * eps * n * cond(A) * max|x| where the oracle finishes in seconds, and
the normwise backward error ||op(A) X - B||_F <= 8 * eps * n * ||A||_F ||X||_F always (the criterion
of the reference's own solve checks, test/lu.cpp:180-199 VERIFY_IS_APPROX(m2, m1 * m3))."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from eigen_amd import _lib
from eigen_amd import lu as ealu
from test_getrs_oracle import CASES, backward_error, factor
from test_lu_oracle import random_matrix

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", range(4))
def test_f77_getrs_vs_oracle(kind):
    dt = ol.DTYPES[kind]
    eps = np.finfo(dt).eps
    for (n, nrhs) in CASES + [(700, 129), (1030, 515)]:
        a, lu, t = factor(dt, n, 7 * n + nrhs)
        cond = np.linalg.cond(a.astype(np.complex128 if kind >= 2 else np.float64))
        for trans in "NTC":
            rng = np.random.default_rng(n + nrhs)
            b0 = random_matrix(rng, (n + 2, nrhs), dt)
            b0[n:] = -1.0e10
            b1, b2 = b0.copy(order="F"), b0.copy(order="F")
            ipiv1, ipiv2 = (t + 1).astype(np.int32), (t + 1).astype(np.int32)
            assert ol.call_f77_getrs(ol.oracle_getrs(kind), trans, n, nrhs, lu, n, ipiv1, b1, n + 2) == 0
            assert ealu.getrs(trans, lu, ipiv2, b2[:n]) == 0
            assert np.array_equal(ipiv2, t + 1)
            assert np.array_equal(b2[n:], b0[n:]), "rows beyond n were written"
            assert np.abs(b2[:n] - b1[:n]).max() <= 16 * eps * n * cond * np.abs(b1[:n]).max(), (n, nrhs, trans)
            assert backward_error(a, trans, b2[:n], b0[:n]) <= 8 * eps * n, (n, nrhs, trans)


@pytest.mark.parametrize("kind", [_lib.F64, _lib.C64])
def test_factor_and_solve_on_the_device(kind):
    """getrf_dev then getrs_dev on device-resident operands, all three forms, at a size with many runs of
    row exchanges (n > 256) and ragged extents; checked by the backward error in float64 on the host."""
    import torch
    from eigen_amd.device import DeviceMatrix
    dt = ol.DTYPES[kind]
    n, nrhs = 3000, 1111
    a = DeviceMatrix(n, n, kind).fill_uniform(11)
    a0 = a.to_numpy().copy()
    transp, nb, fz = ealu.getrf_dev(a)
    assert fz == -1
    for op, trans in ((_lib.OP_N, "N"), (_lib.OP_T, "T"), (_lib.OP_C, "C")):
        b = DeviceMatrix(n, nrhs, kind).fill_uniform(12)
        b0 = b.to_numpy().copy()
        ealu.getrs_dev(op, a, transp, b)
        torch.cuda.synchronize()
        x = b.to_numpy()
        assert backward_error(a0, trans, x, b0) <= 8 * np.finfo(dt).eps * n, trans


def test_solve_at_the_configured_lu_size():
    """32768 x 32768 float64 factors (the configured factorisation), 2048 right-hand sides: sampled
    check of A X = B on 64 rows of A regenerated from the counter-based generator."""
    import torch
    from eigen_amd.device import DeviceMatrix
    n, nrhs = 32768, 2048
    a = DeviceMatrix(n, n, _lib.F64).fill_uniform(21)
    rows = torch.as_tensor(np.random.default_rng(5).choice(n, 64, replace=False), device="cuda")
    a_rows = a.as_strided()[rows].clone()
    a_norm = float(torch.linalg.norm(a.as_strided()))
    transp, nb, fz = ealu.getrf_dev(a)
    assert fz == -1
    b = DeviceMatrix(n, nrhs, _lib.F64).fill_uniform(22)
    b_rows = b.as_strided()[rows].clone()
    ealu.getrs_dev(_lib.OP_N, a, transp, b)
    torch.cuda.synchronize()
    x = b.as_strided()
    res = float(torch.linalg.norm(a_rows @ x - b_rows))
    bound = 8 * np.finfo(np.float64).eps * n * (a_norm * np.sqrt(64 / n)) * float(torch.linalg.norm(x))
    assert res <= bound, (res, bound)
