"""The two drop-in surfaces at the configured sizes, with host pointers:

* `?gemm_` (blas/level3_impl.h:12-76) on 8192-class operands, so that the column-panel
  pipeline of the host path (upload of panel j+1, kernel on panel j, return of panel j-1)
  runs with several panels: leading dimensions larger than the extents, rogue values in the
  padding rows of C (blas/testing/dblat3.f:586-597 checks that they survive), every
  transpose / conjugate form the configuration names for the complex kinds;
* `operator*` on Eigen matrices at 16384^3 through both bindings (EIGEN_USE_BLAS,
  Eigen/src/Core/products/GeneralMatrixMatrix_BLAS.h:56-108, and the header-level seam): the
  check programs compare sampled rows with Eigen's coefficient-based lazyProduct before they
  print a rate.

Tolerance: the Netlib test ratio of sampled entries (recomputed on the host in double
precision from regenerated rows / columns of the operands) below 16, as in test_gemm_gpu.py."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
import rng_np
import eigen_amd
from test_gemm_gpu import ratio
from test_oracle import RATIO_LIMIT

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROGUE = -1.0e10


def host_product_sampled(kind, opa, opb, m, n, k, alpha, beta, pad, nsamp=40):
    dt = ol.DTYPES[kind]
    ar, ac = (m, k) if opa == "N" else (k, m)
    br, bc = (k, n) if opb == "N" else (n, k)
    a = eigen_amd.fill_uniform_host(dt, ar, ac, 1, ld=ar + pad)
    b = eigen_amd.fill_uniform_host(dt, br, bc, 2, ld=br + pad)
    c = eigen_amd.fill_uniform_host(dt, m, n, 3, ld=m + pad)
    c.base[m:, :] = ROGUE
    eigen_amd.f77_gemm(opa, opb, m, n, k, alpha, a.base, ar + pad, b.base, br + pad, beta, c.base, m + pad)
    assert (c.base[m:, :] == ROGUE).all(), "padding rows of C were written"
    rng = np.random.default_rng(m + 7 * n + kind)
    # rows and columns from every column panel and both ends of the matrix
    rows = np.unique(np.concatenate([[0, m - 1], rng.integers(0, m, nsamp)]))
    cols = np.unique(np.concatenate([[0, n - 1, n // 2 - 1, n // 2], rng.integers(0, n, nsamp)]))
    kk = np.arange(k)
    a_rows = rng_np.entries(dt, 1, rows[:, None], kk[None, :]) if opa == "N" else \
        rng_np.entries(dt, 1, kk[None, :], rows[:, None])
    b_cols = rng_np.entries(dt, 2, kk[:, None], cols[None, :]) if opb == "N" else \
        rng_np.entries(dt, 2, cols[None, :], kk[:, None])
    if opa == "C":
        a_rows = a_rows.conj()
    if opb == "C":
        b_cols = b_cols.conj()
    c0 = rng_np.entries(dt, 3, rows[:, None], cols[None, :])
    wide = np.complex128 if kind >= 2 else np.float64
    want = alpha * (a_rows.astype(wide) @ b_cols.astype(wide)) + beta * c0.astype(wide)
    got = c[np.ix_(rows, cols)]
    r = ratio(kind, got, want, alpha, a_rows, b_cols, beta, c0)
    assert r < RATIO_LIMIT, (kind, opa, opb, m, n, k, r)


def test_dgemm_host_pointers_two_panels():
    host_product_sampled(1, "N", "N", 8192, 8192, 8192, 1.0, 1.0, pad=8)


def test_dgemm_host_pointers_ragged_extents():
    host_product_sampled(1, "T", "N", 8200, 8216, 4100, -1.0, 1.0, pad=3)


def test_sgemm_host_pointers_two_panels():
    host_product_sampled(0, "N", "T", 8192, 16384, 4096, 0.7, 1.3, pad=8)


@pytest.mark.parametrize("ops", ["NN", "TN", "NC", "CC"])
def test_zgemm_host_pointers_configured_size(ops):
    host_product_sampled(3, ops[0], ops[1], 8192, 8192, 8192, 0.7 - 0.9j, 1.3 - 1.1j, pad=8)


@pytest.mark.parametrize("ops", ["NN", "CN", "NT"])
def test_cgemm_host_pointers_configured_size(ops):
    host_product_sampled(2, ops[0], ops[1], 8192, 8192, 8192, 0.7 - 0.9j, 1.3 - 1.1j, pad=8)


def test_beta_zero_host_pointers_skips_the_upload_of_c():
    """beta == 0 with NaNs in C at a size with several panels (blas/level3_impl.h:64)."""
    kind, n = 1, 6144
    dt = ol.DTYPES[kind]
    a = eigen_amd.fill_uniform_host(dt, n, n, 1)
    b = eigen_amd.fill_uniform_host(dt, n, n, 2)
    c = np.full((n + 4, n), np.nan, dtype=dt, order="F")
    eigen_amd.f77_gemm("N", "N", n, n, n, 1.0, a, n, b, n, 0.0, c, n + 4)
    assert np.isnan(c[n:]).all() and np.isfinite(c[:n]).all()
    rows = np.array([0, 17, n - 1])
    want = a[rows].astype(np.float64) @ b
    g = np.abs(a[rows]) @ np.abs(b)
    assert (np.abs(c[rows] - want) / (ol.EPS[kind] * g)).max() < RATIO_LIMIT


@pytest.mark.parametrize("binary", ["dropin_blas", "dropin_seam"])
@pytest.mark.parametrize("scalar", ["double", "float"])
def test_operator_star_at_16384(binary, scalar):
    path = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs the reference tree at build time)")
    out = subprocess.run([path, "bench", "16384", scalar], capture_output=True, text=True, timeout=600)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-500:]
    assert "VERIFIED" in out.stdout and "MISMATCH" not in out.stdout
    assert re.search(r"host-to-host: .* TFLOP/s", out.stdout)


def test_dgemm_65536_on_one_gpu_crosses_the_32_bit_element_offset():
    """The configured multi-GPU size on ONE device (3 x 32 GiB resident, one launch of about 8 s): lda = 65536 fits
    the ABI's 32-bit dimensions, but an element's offset reaches 2^32 (Eigen's Index is ptrdiff_t,
    Eigen/src/Core/util/Macros.h:34-36), so every address inside the kernels must be formed in 64 bits.  Parity
    only: the Netlib test ratio of sampled entries, with rows and columns on both sides of the column where the
    offset j * lda passes 2^32 (j = 65536) -- and of the entries right at the 2^31 / 2^32 / 2^32 + 2^31 marks."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    free, _ = torch.cuda.mem_get_info()
    n = 65536
    if free < 3 * n * n * 8 + (4 << 30):
        pytest.skip("needs 100 GiB of free HBM")
    kind = 1
    a = DeviceMatrix(n, n, kind).fill_uniform(1)
    b = DeviceMatrix(n, n, kind).fill_uniform(2)
    c = DeviceMatrix(n, n, kind).fill_uniform(3)
    gemm_dev("N", "N", 1.0, a, b, 1.0, c)
    torch.cuda.synchronize()
    rng = np.random.default_rng(65536)
    # columns on both sides of element offsets 2^30, 2^31 and 2^31 + 2^30 (byte offsets 2^33, 2^34, ...; j = offset / lda)
    marks = np.array([0, 1, 16383, 16384, 32767, 32768, 49151, 49152, n - 2, n - 1])
    rows = np.unique(np.concatenate([marks, rng.integers(0, n, 12)]))
    cols = np.unique(np.concatenate([marks, rng.integers(0, n, 12)]))
    kk = np.arange(n)
    a_rows = rng_np.entries(np.float64, 1, rows[:, None], kk[None, :])
    b_cols = rng_np.entries(np.float64, 2, kk[:, None], cols[None, :])
    c0 = rng_np.entries(np.float64, 3, rows[:, None], cols[None, :])
    want = a_rows @ b_cols + c0
    sub = c.as_strided()[torch.as_tensor(rows, device="cuda")][:, torch.as_tensor(cols, device="cuda")].cpu().numpy()
    r = ratio(kind, sub, want, 1.0, a_rows, b_cols, 1.0, c0)
    assert r < RATIO_LIMIT, r
    # nothing outside the sampled set went astray either: column sums of C against A * (B e) + C0 e would need the host;
    # instead the device's own checksum: e^T C = (e^T A) B + e^T C0, all of it on the device, to a k * eps bound
    ones = torch.ones(n, dtype=torch.float64, device="cuda")
    lhs = ones @ c.as_strided()
    c.fill_uniform(3)
    rhs = (ones @ a.as_strided()) @ b.as_strided() + ones @ c.as_strided()
    scale = (ones @ a.as_strided().abs()) @ b.as_strided().abs() + n
    assert float(((lhs - rhs).abs() / scale).max()) < 64 * n * np.finfo(np.float64).eps
