"""The kernels exist with two tiles per kind (double: 256 x 128, one workgroup per compute unit, and 128 x 64, two
per unit; float: 256 x 256 and 128 x 128; complex double 128 x 128 and 64 x 64; complex float 128 x 256 and 64 x 128)
and the launcher picks one per product.  Either tile, fixed through EIGAMD_<KIND>_TILE, must give the
oracle's product at the Netlib tolerance (test ratio < 16, see test_gemm_gpu.py) on whole-tile and ragged extents
and every operand form; a product that overwrites one of its operands always takes the large tile (the
triangular solves rely on it), whatever the variable says for the others."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from test_gemm_gpu import ratio
from test_oracle import RATIO_LIMIT, opmat

pytestmark = pytest.mark.gpu

SHAPES = [(128, 64, 16), (129, 65, 17), (256, 1024, 256), (1000, 1000, 1000), (300, 200, 1500), (2048, 2048, 256),
          (64, 64, 8192), (1, 1, 1), (127, 63, 4100)]


VARIABLE = {eigen_amd.F64: "EIGAMD_F64_TILE", eigen_amd.F32: "EIGAMD_F32_TILE", eigen_amd.C64: "EIGAMD_C64_TILE",
            eigen_amd.C32: "EIGAMD_C32_TILE"}


@pytest.fixture
def tile_env():
    old = {v: os.environ.get(v) for v in VARIABLE.values()}
    yield
    for v, o in old.items():
        if o is None:
            os.environ.pop(v, None)
        else:
            os.environ[v] = o


@pytest.mark.parametrize("kind", [eigen_amd.F64, eigen_amd.F32, eigen_amd.C64, eigen_amd.C32])
@pytest.mark.parametrize("ops", ["NN", "TN", "NT", "TT", "CN", "NC"])
def test_both_tiles_vs_oracle(kind, ops, tile_env):
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    if kind < 2 and "C" in ops:
        pytest.skip("'C' is 'T' for real kinds")
    opa, opb = ops
    for si, (m, n, k) in enumerate(SHAPES):
        ar, ac = (m, k) if opa == "N" else (k, m)
        br, bc = (k, n) if opb == "N" else (n, k)
        pad = 0 if si % 2 == 0 else 3
        a = DeviceMatrix(ar, ac, kind, ld=ar + pad).fill_uniform(11 + si)
        b = DeviceMatrix(br, bc, kind, ld=br + pad).fill_uniform(23 + si)
        ha, hb = a.to_numpy(), b.to_numpy()
        alpha, beta = [(1.0, 1.0), (0.7, 1.3), (-1.0, 0.0)][si % 3]
        if kind >= 2 and si % 2:
            alpha, beta = alpha * (0.7 - 0.9j), beta * (1.3 - 1.1j)
        got = {}
        for tile in ("1", "2", "0"):
            os.environ[VARIABLE[kind]] = tile
            c = DeviceMatrix(m, n, kind, ld=m + pad).fill_uniform(37 + si)
            hc = c.to_numpy()
            gemm_dev(opa, opb, alpha, a, b, beta, c)
            torch.cuda.synchronize()
            got[tile] = c.to_numpy()
        want = np.array(hc, order="F", copy=True)
        ol.call_f77(ol.oracle_gemm(kind), opa, opb, m, n, k, alpha, np.asfortranarray(ha), ar, np.asfortranarray(hb), br,
                    beta, want, m)
        for tile, g in got.items():
            r = ratio(kind, g, want, alpha, opmat(ha, opa), opmat(hb, opb), beta, hc)
            assert r < RATIO_LIMIT, (tile, ops, m, n, k, r)


@pytest.mark.parametrize("kind", [eigen_amd.F64, eigen_amd.F32, eigen_amd.C64, eigen_amd.C32])
def test_in_place_products_keep_the_large_tile(kind, tile_env):
    """B <- X * B with B 256 rows tall: with the small tile two workgroups per column strip would each read
    all of B's rows and overwrite half of them."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    os.environ[VARIABLE[kind]] = "0"
    eps = np.finfo(np.float64 if kind in (eigen_amd.F64, eigen_amd.C64) else np.float32).eps
    rows = 256 if kind < 2 else 128   # the large tile's extent along m
    for n in (256, 1024, 4096):
        x = DeviceMatrix(rows, rows, kind).fill_uniform(5)
        b = DeviceMatrix(rows, n, kind).fill_uniform(6)
        wide = np.complex128 if kind >= 2 else np.float64
        want = x.to_numpy().astype(wide) @ b.to_numpy().astype(wide)
        gemm_dev("N", "N", 1.0, x, b, 0.0, b)
        torch.cuda.synchronize()
        err = np.abs(b.to_numpy() - want).max()
        assert err <= 256 * 8 * eps, (n, err)
