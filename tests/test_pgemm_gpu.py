"""The multi-device product of the C ABI (eigamd_pgemm_*, eigen_amd/csrc/pgemm.hip) on one GPU:
"virtual grids" put several ranks of the process grid on the same device, so the 2-D
block-cyclic distribution, the panel exchange one panel ahead and the local products run
exactly as they do with one device per rank, except that a panel piece travels by a
device-to-device copy.  Checked against the CPU oracle (test ratio < 16, as in
test_gemm_gpu.py) for every scalar kind, transposed / conjugated operands, extents that are
not multiples of the block, grids with more ranks than blocks; against the single-device
product at a size where every rank multiplies several tiles; the RCCL transport on a grid
of one device (library loading, communicator, grouped broadcast calls)."""
import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from eigen_amd import _lib
from eigen_amd.pgemm import PEER, RCCL, Pgemm, pgemm_host
from test_gemm_gpu import ratio
from test_oracle import RATIO_LIMIT, opmat

pytestmark = pytest.mark.gpu

GRIDS = [(1, 2), (2, 2), (2, 4), (3, 2), (1, 1)]


@pytest.mark.parametrize("kind", range(4))
@pytest.mark.parametrize("grid", GRIDS)
def test_virtual_grid_vs_oracle(kind, grid):
    dt = ol.DTYPES[kind]
    nprow, npcol = grid
    devices = [0] * (nprow * npcol)
    cases = [("N", "N", 300, 200, 500, 64), ("T", "N", 257, 130, 190, 64), ("N", "C", 129, 300, 65, 32),
             ("C", "T", 70, 90, 400, 48), ("N", "N", 40, 30, 20, 64), ("T", "T", 200, 1, 333, 16)]
    for ci, (opa, opb, m, n, k, nb) in enumerate(cases):
        ar, ac = (m, k) if opa == "N" else (k, m)
        br, bc = (k, n) if opb == "N" else (n, k)
        pad = ci % 3
        a = eigen_amd.fill_uniform_host(dt, ar, ac, 11 + ci, ld=ar + pad)
        b = eigen_amd.fill_uniform_host(dt, br, bc, 23 + ci, ld=br + pad)
        c0 = eigen_amd.fill_uniform_host(dt, m, n, 37 + ci, ld=m + pad)
        c0.base[m:, :] = -1.0e10
        alpha, beta = [(1.0, 1.0), (0.7, 1.3), (-1.0, 0.0)][ci % 3]
        if kind >= 2:
            alpha, beta = alpha * (0.7 - 0.9j), beta * (1.3 - 1.1j)
        want = np.array(c0.base, order="F", copy=True)
        ol.call_f77(ol.oracle_gemm(kind), opa, opb, m, n, k, alpha, a.base, ar + pad, b.base, br + pad, beta, want, m + pad)
        got = np.array(c0.base, order="F", copy=True)
        pgemm_host(opa, opb, alpha, a, b, beta, got[:m], nprow, npcol, nb, devices=devices)
        assert np.array_equal(got[m:], c0.base[m:]), "padding rows of C were written"
        r = ratio(kind, got[:m], want[:m], alpha, opmat(a, opa), opmat(b, opb), beta, c0)
        assert r < RATIO_LIMIT, (kind, grid, opa, opb, m, n, k, nb, r)


def test_k_zero_scales_c_on_every_rank():
    c0 = eigen_amd.fill_uniform_host(np.float64, 150, 90, 5)
    a = np.zeros((150, 0), order="F")
    b = np.zeros((0, 90), order="F")
    got = c0.copy(order="F")
    pgemm_host("N", "N", 1.0, a, b, 1.3, got, 2, 2, 32, devices=[0] * 4)
    assert np.allclose(got, 1.3 * c0, rtol=1e-15, atol=0)


def test_device_resident_operands_match_the_single_device_fill_and_product():
    """Handle form: the synthetic operands generated on the grid equal the single-device ones
    element for element; repeated products accumulate like the single-device product."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    kind, n, nb = 1, 3072, 512
    g = Pgemm(kind, "N", "T", n, n, n, 2, 4, nb, devices=[0] * 8)
    g.fill("A", 1).fill("B", 2).fill("C", 3)
    a = DeviceMatrix(n, n, kind).fill_uniform(1)
    b = DeviceMatrix(n, n, kind).fill_uniform(2)
    c = DeviceMatrix(n, n, kind).fill_uniform(3)
    g.sync()
    assert np.array_equal(g.gather("A"), a.to_numpy()) and np.array_equal(g.gather("B"), b.to_numpy())
    for _ in range(2):
        g.run(1.0, 1.0)
        gemm_dev("N", "T", 1.0, a, b, 1.0, c)
    g.sync()
    torch.cuda.synchronize()
    got, want = g.gather("C"), c.to_numpy()
    gauge = 2 * (a.as_strided().abs() @ b.as_strided().abs().T).cpu().numpy() + 1.0
    assert (np.abs(got - want) / (ol.EPS[kind] * gauge)).max() < RATIO_LIMIT
    g.close()


def test_rccl_transport_on_a_grid_of_one():
    """Loads librccl, creates the row / column communicators and issues the grouped broadcast
    calls; with one device there is nothing to receive, the product must still be right."""
    kind, m, n, k = 1, 200, 150, 300
    a = eigen_amd.fill_uniform_host(np.float64, m, k, 1)
    b = eigen_amd.fill_uniform_host(np.float64, k, n, 2)
    c0 = eigen_amd.fill_uniform_host(np.float64, m, n, 3)
    want = c0.copy(order="F")
    ol.call_f77(ol.oracle_gemm(kind), "N", "N", m, n, k, 0.7, a, m, b, k, 1.3, want, m)
    got = c0.copy(order="F")
    pgemm_host("N", "N", 0.7, a, b, 1.3, got, 1, 1, 64, devices=[0], transport=RCCL)
    assert ratio(kind, got, want, 0.7, a, b, 1.3, c0) < RATIO_LIMIT


def test_arguments():
    a = np.zeros((4, 4), order="F")
    with pytest.raises(_lib.EigenAmdError):
        pgemm_host("N", "N", 1.0, a, a, 0.0, a.copy(order="F"), 1, 2, 2, devices=[0, 0], transport=RCCL)   # RCCL: one device per rank
    with pytest.raises(_lib.EigenAmdError):
        pgemm_host("N", "N", 1.0, a, a, 0.0, a.copy(order="F"), 1, 2, 2, devices=[0, 99])


def test_host_entry_deals_a_large_product_over_the_grid():
    """?gemm_ itself goes through the grid when several devices are visible and the product is
    large; here four ranks on the one device stand in for four devices."""
    from test_lu_config_gpu import env
    from test_gemm_config_gpu import host_product_sampled
    with env(EIGAMD_MULTI_GPU_VIRTUAL=4, EIGAMD_MULTI_GPU_NB=512, EIGAMD_MULTI_GPU_MIN_WORK=1e9):
        host_product_sampled(1, "N", "N", 4096, 4200, 3000, 1.0, 1.0, pad=8)
        host_product_sampled(3, "C", "N", 2048, 2100, 1500, 0.7 - 0.9j, 1.3 - 1.1j, pad=3)


def test_beta_zero_never_reads_c():
    """beta == 0: C is not uploaded at all (blas/level3_impl.h:64 overwrites without reading), so NaNs in the
    caller's C must not reach the result."""
    kind, m, n, k = 1, 700, 600, 900
    a = eigen_amd.fill_uniform_host(np.float64, m, k, 1)
    b = eigen_amd.fill_uniform_host(np.float64, k, n, 2)
    want = np.zeros((m, n), order="F")
    ol.call_f77(ol.oracle_gemm(kind), "N", "N", m, n, k, 0.7, a, m, b, k, 0.0, want, m)
    got = np.full((m, n), np.nan, order="F")
    pgemm_host("N", "N", 0.7, a, b, 0.0, got, 2, 2, 128, devices=[0] * 4)
    assert ratio(kind, got, want, 0.7, a, b, 0.0, np.zeros((m, n), order="F")) < RATIO_LIMIT
