"""Parity of the HIP GEMM path (through the C ABI) with the CPU oracle, the
golden outputs of the reference, and the Netlib known-answer driver.

Tolerance (floating point): the Netlib test ratio
    max_ij |c_ij - ref_ij| / (eps * (abs1(alpha) * sum_l |a_il||b_lj| + abs1(beta) |c0_ij|))
must stay below 16, the THRESHOLD of blas/testing/?blat3.dat (eps = machine
epsilon of the kind, abs1(z) = |Re z| + |Im z|).  This is a componentwise form of
the normwise bound k * eps * |A| * |B|; observed values are 0.4-4.
"""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import rng_np
import eigen_amd
from eigen_amd import _lib
from test_oracle import GOLDEN, NETLIB, RATIO_LIMIT, case_ratio, golden, golden_operands, opmat

pytestmark = pytest.mark.gpu


def abs1(z):
    return np.abs(z.real) + np.abs(z.imag) if np.iscomplexobj(z) else np.abs(z)


def ratio(kind, got, want, alpha, a_op, b_op, beta, c0):
    g = abs1(np.asarray(alpha)) * (abs1(a_op).astype(np.float64) @ abs1(b_op).astype(np.float64))
    g = g + abs1(np.asarray(beta)) * abs1(c0)
    g = np.where(g == 0, 1.0, g)
    d = got.astype(np.complex128) - want.astype(np.complex128)
    return float(((np.abs(d.real) + np.abs(d.imag)) / (ol.EPS[kind] * g)).max()) if got.size else 0.0


def test_device_present():
    assert _lib.lib.eigamd_device_ok() == 1, _lib.lib.eigamd_last_error()


@pytest.mark.parametrize("kind", range(4))
def test_netlib_known_answers_gpu(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    fn = getattr(_lib.lib, ol.SUFFIX[kind] + "gemm_")
    rep = ol.run_netlib_chk1(kind, fn, p)
    assert rep.ncalls == 17496
    assert not rep.fatal and rep.nfail_same == 0 and rep.nfail_ratio == 0 and rep.nfail_xerbla == 0
    assert rep.errmax < p["thresh"]


def test_golden_reference_outputs():
    z, meta = golden()
    for idx, case in enumerate(meta["cases"]):
        a, b, c0 = golden_operands(case, meta["seeds"])
        m, n, k, pad, s = case["m"], case["n"], case["k"], case["pad"], case["stride"]
        alpha, beta = complex(*case["alpha"]), complex(*case["beta"])
        if case["kind"] < 2:
            alpha, beta = alpha.real, beta.real
        cbuf = np.array(c0.base, order="F", copy=True)
        if "J" in (case["opa"], case["opb"]):
            eigen_amd.gemm_host(case["opa"], case["opb"], m, n, k, alpha, a.base, a.shape[0] + pad, b.base,
                                b.shape[0] + pad, beta, cbuf, m + pad)
        else:
            eigen_amd.f77_gemm(case["opa"], case["opb"], m, n, k, alpha, a.base, a.shape[0] + pad, b.base,
                               b.shape[0] + pad, beta, cbuf, m + pad)
        assert np.array_equal(cbuf[m:, :], c0.base[m:, :]), "C padding rows were written"
        got = cbuf[:m:s, :n:s]
        r = case_ratio(case, got, z[f"opstar_{idx}"], a, b, c0)
        if f"f77_{idx}" in z.files:
            r = max(r, case_ratio(case, got, z[f"f77_{idx}"], a, b, c0))
        assert r < RATIO_LIMIT, (idx, case, r)


SHAPES = [(1, 1, 1), (1, 200, 33), (200, 1, 33), (7, 9, 1), (127, 129, 31), (128, 128, 128), (129, 127, 130),
          (255, 257, 65), (384, 256, 96), (300, 517, 211), (1000, 40, 77), (513, 515, 1030)]


@pytest.mark.parametrize("kind", range(4))
@pytest.mark.parametrize("ops", ["NN", "TN", "NT", "TT", "CN", "NC", "CC", "CT", "TC"])
def test_host_path_vs_oracle(kind, ops):
    if kind < 2 and "C" in ops and ops not in ("CN", "NC"):
        pytest.skip("'C' is 'T' for real kinds; two forms suffice")
    dt = ol.DTYPES[kind]
    opa, opb = ops
    rng = np.random.default_rng(kind * 100 + hash(ops) % 97)
    for si, (m, n, k) in enumerate(SHAPES):
        pad = si % 4
        ar, ac = (m, k) if opa == "N" else (k, m)
        br, bc = (k, n) if opb == "N" else (n, k)
        a = eigen_amd.fill_uniform_host(dt, ar, ac, 11 + si, ld=ar + pad)
        b = eigen_amd.fill_uniform_host(dt, br, bc, 23 + si, ld=br + pad)
        c0 = eigen_amd.fill_uniform_host(dt, m, n, 37 + si, ld=m + pad)
        c0.base[m:, :] = -1.0e10  # rogue padding rows, as ?MAKE writes them
        alpha, beta = [(1.0, 1.0), (0.7, 1.3), (-1.0, 1.0), (0.7, 0.0), (0.0, 1.3)][si % 5]
        if kind >= 2 and si % 2:
            alpha, beta = alpha * (0.7 - 0.9j), beta * (1.3 - 1.1j)
        want = np.array(c0.base, order="F", copy=True)
        ol.call_f77(ol.oracle_gemm(kind), opa, opb, m, n, k, alpha, a.base, ar + pad, b.base, br + pad, beta, want,
                    m + pad)
        got = np.array(c0.base, order="F", copy=True)
        eigen_amd.f77_gemm(opa, opb, m, n, k, alpha, a.base, ar + pad, b.base, br + pad, beta, got, m + pad)
        assert np.array_equal(got[m:, :], c0.base[m:, :]), "C padding rows were written"
        r = ratio(kind, got[:m], want[:m], alpha, opmat(a, opa), opmat(b, opb), beta, c0)
        assert r < RATIO_LIMIT, (kind, ops, m, n, k, r)


@pytest.mark.parametrize("kind", range(4))
def test_beta_zero_ignores_nan_in_c(kind):
    dt = ol.DTYPES[kind]
    a = eigen_amd.fill_uniform_host(dt, 70, 50, 1)
    b = eigen_amd.fill_uniform_host(dt, 50, 90, 2)
    c = np.full((70, 90), np.nan, dtype=dt, order="F")
    eigen_amd.gemm(1.0, a, b, 0.0, c)
    assert np.isfinite(c).all()
    want = np.zeros((70, 90), dtype=dt, order="F")
    ol.call_f77(ol.oracle_gemm(kind), "N", "N", 70, 90, 50, 1.0, a, 70, b, 50, 0.0, want, 70)
    assert ratio(kind, c, want, 1.0, a, b, 0.0, want) < RATIO_LIMIT


@pytest.mark.parametrize("kind", range(4))
def test_k_zero_scales_c(kind):
    dt = ol.DTYPES[kind]
    c0 = eigen_amd.fill_uniform_host(dt, 33, 21, 5, ld=35)
    a = np.zeros((33, 1), dtype=dt, order="F")
    for beta in (0.0, 1.0, 1.3):
        got = np.array(c0.base, order="F", copy=True)
        want = np.array(c0.base, order="F", copy=True)
        eigen_amd.f77_gemm("N", "N", 33, 21, 0, 1.0, a, 33, a, 1, beta, got, 35)
        ol.call_f77(ol.oracle_gemm(kind), "N", "N", 33, 21, 0, 1.0, a, 33, a, 1, beta, want, 35)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("kind", range(4))
def test_device_entry_blocks_and_unaligned_views(kind):
    """Device-pointer entry on sub-blocks of larger matrices (the LU trailing-update
    form) including views whose base is not 16-byte aligned."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    dt = ol.DTYPES[kind]
    big_a = DeviceMatrix(700, 400, kind, ld=704).fill_uniform(1)
    big_b = DeviceMatrix(400, 600, kind, ld=400).fill_uniform(2)
    big_c = DeviceMatrix(700, 600, kind, ld=712).fill_uniform(3)
    ha, hb, hc = big_a.to_numpy(), big_b.to_numpy(), big_c.to_numpy()
    assert np.array_equal(ha, rng_np.matrix(dt, 1, 700, 400))  # device and host generators agree
    for (r0, c0, m, n, k0, k) in [(0, 0, 512, 384, 0, 256), (128, 128, 300, 260, 16, 129), (1, 3, 257, 255, 5, 77)]:
        a = big_a.block(r0, k0, m, k)
        b = big_b.block(k0, c0, k, n)
        c = big_c.block(r0, c0, m, n)
        before = big_c.to_numpy()
        gemm_dev("N", "N", -1.0, a, b, 1.0, c)
        torch.cuda.synchronize()
        after = big_c.to_numpy()
        want = before[r0:r0 + m, c0:c0 + n].copy(order="F")
        ol.call_f77(ol.oracle_gemm(kind), "N", "N", m, n, k, -1.0, np.asfortranarray(ha[r0:r0 + m, k0:k0 + k]), m,
                    np.asfortranarray(hb[k0:k0 + k, c0:c0 + n]), k, 1.0, want, m)
        r = ratio(kind, after[r0:r0 + m, c0:c0 + n], want, -1.0, ha[r0:r0 + m, k0:k0 + k], hb[k0:k0 + k, c0:c0 + n],
                  1.0, before[r0:r0 + m, c0:c0 + n])
        assert r < RATIO_LIMIT, (kind, r0, c0, m, n, k, r)
        mask = np.ones_like(before, dtype=bool)
        mask[r0:r0 + m, c0:c0 + n] = False
        assert np.array_equal(after[mask], before[mask]), "wrote outside the C block"


def _sampled_check(kind, opa, opb, n, alpha, beta, nsamp=48):
    """Full-size product on the device; a sample of rows x columns of C is recomputed
    on the host from regenerated rows/columns of the operands (float64 accumulate)."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    dt = ol.DTYPES[kind]
    a = DeviceMatrix(n, n, kind).fill_uniform(1)
    b = DeviceMatrix(n, n, kind).fill_uniform(2)
    c = DeviceMatrix(n, n, kind).fill_uniform(3)
    gemm_dev(opa, opb, alpha, a, b, beta, c)
    torch.cuda.synchronize()
    rng = np.random.default_rng(n + kind)
    rows = np.unique(np.concatenate([[0, n - 1], rng.integers(0, n, nsamp)]))
    cols = np.unique(np.concatenate([[0, n - 1], rng.integers(0, n, nsamp)]))
    kk = np.arange(n)
    a_rows = rng_np.entries(dt, 1, rows[:, None], kk[None, :]) if opa in "NJ" else \
        rng_np.entries(dt, 1, kk[None, :], rows[:, None])          # op(A)[rows, :]
    b_cols = rng_np.entries(dt, 2, kk[:, None], cols[None, :]) if opb in "NJ" else \
        rng_np.entries(dt, 2, cols[None, :], kk[:, None])          # op(B)[:, cols]
    if opa in "CJ":
        a_rows = a_rows.conj()
    if opb in "CJ":
        b_cols = b_cols.conj()
    c0 = rng_np.entries(dt, 3, rows[:, None], cols[None, :])
    wide = np.complex128 if kind >= 2 else np.float64
    want = alpha * (a_rows.astype(wide) @ b_cols.astype(wide)) + beta * c0.astype(wide)
    got = c.as_strided()[torch.as_tensor(rows, device="cuda")][:, torch.as_tensor(cols, device="cuda")].cpu().numpy()
    r = ratio(kind, got, want, alpha, a_rows, b_cols, beta, c0)
    assert r < RATIO_LIMIT, (kind, opa, opb, n, r)
    return a, b, c


@pytest.mark.parametrize("kind,n", [(1, 16384), (0, 16384)])
def test_headline_size_sampled_and_checksum(kind, n):
    """BASELINE configuration (16384^3, C += A*B): sampled entries against the host,
    plus a whole-matrix property: (A*B + C0) * v == A * (B * v) + C0 * v."""
    import torch
    a, b, c = _sampled_check(kind, "N", "N", n, 1.0, 1.0)
    v = torch.linspace(-1, 1, n, dtype=torch.float64, device="cuda")
    A, B, C = a.as_strided().double(), None, c.as_strided().double()
    lhs = C @ v
    del C
    bv = b.as_strided().double() @ v
    from eigen_amd.device import DeviceMatrix
    c0 = DeviceMatrix(n, n, kind).fill_uniform(3)
    rhs = A @ bv + c0.as_strided().double() @ v
    scale = (A.abs() @ (b.as_strided().double().abs() @ v.abs())).max().item()
    err = (lhs - rhs).abs().max().item()
    assert err <= 64 * ol.EPS[kind] * scale, (err, scale)


@pytest.mark.parametrize("kind,ops", [(3, "NN"), (3, "CN"), (3, "NC"), (2, "TN"), (2, "CC"), (2, "NN")])
def test_complex_8192_variants_sampled(kind, ops):
    _sampled_check(kind, ops[0], ops[1], 8192, 0.7 - 0.9j, 1.3 - 1.1j)


@pytest.mark.parametrize("kind", [0, 1])
def test_linearity_at_full_tile_sizes(kind):
    """alpha-linearity on the device path: gemm(2*alpha) - 2*gemm(alpha) == 0 exactly
    (scaling by a power of two commutes with rounding)."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    n = 2048
    a = DeviceMatrix(n, n, kind).fill_uniform(1)
    b = DeviceMatrix(n, n, kind).fill_uniform(2)
    c1 = DeviceMatrix(n, n, kind)
    c2 = DeviceMatrix(n, n, kind)
    gemm_dev("N", "N", 0.5, a, b, 0.0, c1)
    gemm_dev("N", "N", 1.0, a, b, 0.0, c2)
    torch.cuda.synchronize()
    assert torch.equal(c1.as_strided() * 2, c2.as_strided())
