"""eigamd_gemm_dev_noalloc (the entry point that allocates nothing: partial products of a split
reduction in a fixed device array of the library) against eigamd_gemm_dev on the same
operands: identical bits, on two calls in a row, the second with a smaller input; the size
limit; and agreement with the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from eigen_amd import _lib
from test_gemm_gpu import ratio
from test_oracle import RATIO_LIMIT

pytestmark = pytest.mark.gpu


def both(kind, opa, opb, m, n, k, alpha, beta, seed):
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    ar, ac = (m, k) if opa in "NJ" else (k, m)
    br, bc = (k, n) if opb in "NJ" else (n, k)
    a = DeviceMatrix(ar, ac, kind).fill_uniform(seed)
    b = DeviceMatrix(br, bc, kind).fill_uniform(seed + 1)
    c1 = DeviceMatrix(m, n, kind, ld=m + 3).fill_uniform(seed + 2)
    c2 = DeviceMatrix(m, n, kind, ld=m + 3).fill_uniform(seed + 2)
    c0 = c1.to_numpy()
    gemm_dev(opa, opb, alpha, a, b, beta, c1)
    gemm_dev(opa, opb, alpha, a, b, beta, c2, noalloc=True)
    torch.cuda.synchronize()
    return a.to_numpy(), b.to_numpy(), c0, c1.to_numpy(), c2.to_numpy()


@pytest.mark.parametrize("kind", range(4))
def test_two_calls_in_a_row_second_smaller(kind):
    # reductions long enough that both products are split (few tiles, k >= 8 k-tiles)
    for (m, n, k, opa, opb) in [(1000, 520, 4096, "N", "N"), (300, 200, 2048, "T", "N")]:
        alpha, beta = (0.7 - 0.9j, 1.3 - 1.1j) if kind >= 2 else (0.7, 1.3)
        a, b, c0, want, got = both(kind, opa, opb, m, n, k, alpha, beta, 11 + m)
        assert np.array_equal(got, want), (kind, m, n, k)
        # and both agree with the oracle
        ref = c0.copy(order="F")
        ol.call_f77(ol.oracle_gemm(kind), opa, opb, m, n, k, alpha, a, a.shape[0], b, b.shape[0], beta, ref, m)
        a_op = a if opa == "N" else a.T
        assert ratio(kind, got, ref, alpha, a_op, b, beta, c0) < RATIO_LIMIT


def test_unsplit_and_ragged_products():
    for (m, n, k) in [(129, 127, 30), (1, 1, 1), (513, 515, 100), (255, 1030, 9000)]:
        a, b, c0, want, got = both(1, "N", "N", m, n, k, -1.0, 1.0, 5)
        assert np.array_equal(got, want), (m, n, k)
    a, b, c0, want, got = both(0, "N", "N", 700, 300, 0, 1.0, 0.0, 5)   # k = 0, beta = 0: C <- 0
    assert np.array_equal(got, want) and not got.any()


def test_size_limit_is_reported():
    from eigen_amd.device import DeviceMatrix, gemm_dev
    n = 1100                                           # 1100^2 doubles = 9.7 MB > 8 MiB
    a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(1)
    c = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(2)
    before = c.to_numpy()
    with pytest.raises(_lib.EigenAmdError) as e:
        gemm_dev("N", "N", 1.0, a, a, 1.0, c, noalloc=True)
    assert e.value.code == _lib.ERR_ARG and "exceeds the maximum" in str(e.value)
    assert np.array_equal(c.to_numpy(), before)
    gemm_dev("N", "N", 1.0, a.block(0, 0, 1024, n), a.block(0, 0, n, 1024), 1.0, c.block(0, 0, 1024, 1024), noalloc=True)  # exactly 8 MiB
