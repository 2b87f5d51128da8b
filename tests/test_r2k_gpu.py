"""GPU parity tests of the rank-2k updates (?syr2k_, ?her2k_ and eigamd_syr2k_dev) through the C
ABI: the restated Netlib ?CHK5 / ?CHKE rows with the reference's parameter files, golden results
of Eigen's entry points (return values included), the CPU oracle on device-resident operands, and
at 4096 x 2048 the equality of the triangle with two launches of the library's own GEMM.
Tolerance: componentwise |c - c_ref| / (eps (|alpha| (|A||B|^T + |B||A|^T) + |beta||C0|)) < 16."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as ol
from eigen_amd import _lib
from eigen_amd.device import DeviceMatrix, gemm_dev, syr2k_dev
from test_r2k_oracle import FORMS, R2K_CALLS, RATIO_LIMIT, golden, r2k_ratio, run_case, scalars

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")


def f77(kind, herm):
    return getattr(_lib.lib, ol.SUFFIX[kind] + ("her2k_" if herm else "syr2k_"))


@pytest.mark.parametrize("kind,herm", FORMS)
def test_netlib_r2k_known_answers_gpu(kind, herm):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    rep = ol.run_netlib_chk5(kind, f77(kind, herm), herm, p)
    assert rep.ncalls == R2K_CALLS[kind]
    assert not rep.fatal and rep.nfail_same == 0 and rep.nfail_ratio == 0 and rep.nfail_xerbla == 0
    assert rep.errmax < p["thresh"] == RATIO_LIMIT
    assert ol.run_netlib_chke_r2k(kind, f77(kind, herm), herm) == 0


def test_golden_reference_results():
    z, meta = golden()
    for idx, c in enumerate(meta["cases"]):
        a, b, c0, got, ret = run_case(f77(c["kind"], c["herm"]), c)
        assert ret == c["ret"], (idx, c)
        assert r2k_ratio(c, a, b, c0, got, z[f"c{idx}"], c["stride"]) < RATIO_LIMIT, (idx, c)


@pytest.mark.parametrize("kind,herm", FORMS)
def test_device_entry_vs_oracle(kind, herm):
    seed = 900
    for uplo in "UL":
        for trans in ("NC" if herm else "NT"):
            for (n, k, alpha, beta) in [(700, 90, 0.7, 1.3), (513, 300, -1.0, 0.0), (300, 0, 1.0, 0.5), (40, 1000, 1.0, 1.0)]:
                seed += 1
                al = [alpha, -0.5 * alpha if kind >= 2 else 0.0]
                be = [beta, 0.25 * beta if (kind >= 2 and not herm) else 0.0]
                c = dict(kind=kind, herm=herm, uplo=uplo, trans=trans, n=n, k=k, alpha=al, beta=be, pad=seed % 3, seed=seed)
                a, b, c0, want, _ = run_case(ol.oracle_r2k(kind, herm), c)
                ld = max(a.shape[0], 1) + seed % 3
                da = DeviceMatrix.from_numpy(np.array(a), ld=ld)
                db = DeviceMatrix.from_numpy(np.array(b), ld=ld + 1)
                dc = DeviceMatrix.from_numpy(np.array(c0), ld=n + 1)
                alpha_s, beta_s = scalars(c)
                syr2k_dev(uplo, trans, alpha_s, da, db, beta_s, dc, hermitian=herm)
                got = dc.to_numpy()
                assert r2k_ratio(c, a, b, c0, got, want) < RATIO_LIMIT, c


@pytest.mark.parametrize("kind,herm,uplo,trans", [(1, False, "L", "N"), (0, False, "U", "T"), (3, True, "L", "N"), (2, True, "U", "C")])
def test_large_update_matches_gemm_triangle(kind, herm, uplo, trans):
    n, k = 4096, 2048
    alpha, beta = (0.7, 1.3) if kind < 2 else (0.7 - 0.9j, 1.3)
    rows, cols = (n, k) if trans == "N" else (k, n)
    a = DeviceMatrix(rows, cols, kind).fill_uniform(31)
    b = DeviceMatrix(rows, cols, kind).fill_uniform(33)
    c = DeviceMatrix(n, n, kind).fill_uniform(32)
    full = DeviceMatrix(n, n, kind).fill_uniform(32)
    c0 = c.as_strided().clone()
    syr2k_dev(uplo, trans, alpha, a, b, beta, c, hermitian=herm)
    t = "C" if herm else "T"
    alpha2 = np.conj(alpha) if herm else alpha
    if trans == "N":
        gemm_dev("N", t, alpha, a, b, beta, full)
        gemm_dev("N", t, alpha2, b, a, 1.0, full)
    else:
        gemm_dev(t, "N", alpha, a, b, beta, full)
        gemm_dev(t, "N", alpha2, b, a, 1.0, full)
    torch.cuda.synchronize()
    got, ref = c.as_strided(), full.as_strided()
    tri = torch.triu(torch.ones(n, n, dtype=torch.bool, device="cuda")) if uplo == "U" else \
        torch.tril(torch.ones(n, n, dtype=torch.bool, device="cuda"))
    assert torch.equal(got[~tri], c0[~tri])
    strict = tri & ~torch.eye(n, dtype=torch.bool, device="cuda")
    err = (got[strict] - ref[strict]).abs().max().item()
    assert err <= 40 * k * ol.EPS[kind] * abs(alpha), err
    if herm:
        assert torch.all(torch.diagonal(got).imag == 0)
