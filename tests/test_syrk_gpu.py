"""GPU parity tests of the symmetric rank-k update (?syrk_ and eigamd_syrk_dev) through the C
ABI: the restated Netlib ?CHK4 / ?CHKE rows with the reference's parameter files, the golden
results of Eigen's ?syrk_, the CPU oracle on device-resident operands, and at 8192 x 4096 the
equality of the triangle with the library's own (separately verified) GEMM.
Tolerance: componentwise |c - c_ref| / (eps (|alpha| |A||A|^T + |beta||C0|)) < 16."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_lib as ol
import eigen_amd
from eigen_amd import _lib
from eigen_amd.device import DeviceMatrix, gemm_dev, syrk_dev
from test_syrk_oracle import RATIO_LIMIT, SYRK_CALLS, golden, run_case, syrk_ratio

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")


def f77(kind):
    return getattr(_lib.lib, ol.SUFFIX[kind] + "syrk_")


@pytest.mark.parametrize("kind", range(4))
def test_netlib_syrk_known_answers_gpu(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    rep = ol.run_netlib_chk4_syrk(kind, f77(kind), p)
    assert rep.ncalls == SYRK_CALLS[kind]
    assert not rep.fatal and rep.nfail_same == 0 and rep.nfail_ratio == 0 and rep.nfail_xerbla == 0
    assert rep.errmax < p["thresh"] == RATIO_LIMIT
    assert ol.run_netlib_chke_syrk(kind, f77(kind)) == 0


def test_golden_reference_results():
    z, meta = golden()
    for idx, c in enumerate(meta["cases"]):
        a, c0, got, alpha, beta = run_case(f77(c["kind"]), c)
        s = c["stride"]
        want_s = z[f"c{idx}"]
        abs1 = (lambda v: np.abs(v.real) + np.abs(v.imag)) if c["kind"] >= 2 else np.abs
        aa = abs1(np.asarray(a)).astype(np.float64)
        g = abs1(np.asarray(alpha)) * (aa @ aa.T if c["trans"] == "N" else aa.T @ aa) + abs1(np.asarray(beta)) * abs1(c0)
        g = np.where(g == 0, 1.0, g)[::s, ::s]
        n = c["n"]
        i, j = np.indices((n, n))
        tri = ((i <= j) if c["uplo"] == "U" else (i >= j))[::s, ::s]
        gs = got[::s, ::s]
        assert np.array_equal(gs[~tri], want_s[~tri]), (idx, c)
        d = abs1(gs.astype(np.complex128) - want_s.astype(np.complex128)) / (ol.EPS[c["kind"]] * g)
        assert d[tri].max() < RATIO_LIMIT, (idx, c)


@pytest.mark.parametrize("kind", range(4))
def test_device_entry_vs_oracle(kind):
    """Sizes above the 256-wide leaf (so the recursion, the plain off-diagonal products and
    the predicated diagonal leaves all run), odd leading dimensions, k = 0 and beta = 0."""
    seed = 300
    for uplo in "UL":
        for trans in "NT":
            for (n, k, alpha, beta) in [(700, 90, 0.7, 1.3), (513, 300, -1.0, 0.0), (300, 0, 1.0, 0.5), (40, 1000, 1.0, 1.0)]:
                seed += 1
                if kind >= 2:
                    alpha, beta = alpha * (1 - 0.5j), beta * (1 + 0.25j)
                a, c = ol.syrk_operands(kind, uplo, trans, n, k, seed, pad=seed % 3)
                c0 = np.array(c)
                want = np.array(c, order="F")
                wbuf = np.asfortranarray(want)
                ol.call_f77_syrk(ol.oracle_syrk(kind), uplo, trans, n, k, alpha, a, max(a.shape[0], 1) + seed % 3, beta, wbuf, n)
                da = DeviceMatrix.from_numpy(np.array(a), ld=max(a.shape[0], 1) + seed % 3)
                dc = DeviceMatrix.from_numpy(np.array(c), ld=n + 1)
                syrk_dev(uplo, trans, alpha, da, beta, dc)
                got = dc.to_numpy()
                assert syrk_ratio(kind, uplo, trans, alpha, beta, a, c0, got, wbuf) < RATIO_LIMIT, (uplo, trans, n, k)


@pytest.mark.parametrize("kind,uplo,trans", [(1, "L", "N"), (1, "U", "T"), (0, "L", "T"), (3, "U", "N")])
def test_large_update_matches_gemm_triangle(kind, uplo, trans):
    n, k = 8192, 4096
    alpha, beta = (0.7, 1.3) if kind < 2 else (0.7 - 0.9j, 1.3 - 1.1j)
    a = DeviceMatrix(n if trans == "N" else k, k if trans == "N" else n, kind).fill_uniform(11)
    c = DeviceMatrix(n, n, kind).fill_uniform(12)
    full = DeviceMatrix(n, n, kind).fill_uniform(12)
    c0 = c.as_strided().clone()
    syrk_dev(uplo, trans, alpha, a, beta, c)
    if trans == "N":
        gemm_dev("N", "T", alpha, a, a, beta, full)
    else:
        gemm_dev("T", "N", alpha, a, a, beta, full)
    torch.cuda.synchronize()
    got, ref = c.as_strided(), full.as_strided()
    tri = torch.triu(torch.ones(n, n, dtype=torch.bool, device="cuda")) if uplo == "U" else \
        torch.tril(torch.ones(n, n, dtype=torch.bool, device="cuda"))
    assert torch.equal(got[~tri], c0[~tri])                       # the other triangle is untouched
    # same k-order per element in both kernels, up to the tile the element falls in
    err = (got[tri] - ref[tri]).abs().max().item()
    assert err <= 20 * k * ol.EPS[kind] * abs(alpha), err   # |a_ij| < 1: far above sqrt(k) eps rounding noise, far below a wrong term
