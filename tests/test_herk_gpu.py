"""GPU parity tests of the Hermitian rank-k update (cherk_ / zherk_ and eigamd_herk_dev) through
the C ABI: the restated HERK rows of the Netlib ZCHK4 / ZCHKE with the reference's parameter
files, golden results of Eigen's ?herk_, the CPU oracle on device-resident operands, and at
4096 x 2048 the equality of the triangle with the library's own GEMM.
Tolerance: componentwise |c - c_ref| / (eps (|alpha| |A||A|^H + |beta||C0|)) < 16."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as ol
from eigen_amd import _lib
from eigen_amd.device import DeviceMatrix, gemm_dev, herk_dev
from test_herk_oracle import HERK_CALLS, RATIO_LIMIT, golden, herk_ratio, run_case

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")


def f77(kind):
    return getattr(_lib.lib, ol.SUFFIX[kind] + "herk_")


@pytest.mark.parametrize("kind", [2, 3])
def test_netlib_herk_known_answers_gpu(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    rep = ol.run_netlib_chk4_herk(kind, f77(kind), p)
    assert rep.ncalls == HERK_CALLS
    assert not rep.fatal and rep.nfail_same == 0 and rep.nfail_ratio == 0 and rep.nfail_xerbla == 0
    assert rep.errmax < p["thresh"] == RATIO_LIMIT
    assert ol.run_netlib_chke_herk(kind, f77(kind)) == 0


def test_golden_reference_results():
    z, meta = golden()
    for idx, c in enumerate(meta["cases"]):
        a, c0, got = run_case(f77(c["kind"]), c)
        assert herk_ratio(c, a, c0, got, z[f"c{idx}"], c["stride"]) < RATIO_LIMIT, (idx, c)


@pytest.mark.parametrize("kind", [2, 3])
def test_device_entry_vs_oracle(kind):
    seed = 700
    for uplo in "UL":
        for trans in "NC":
            for (n, k, alpha, beta) in [(700, 90, 0.7, 1.3), (513, 300, -1.0, 0.0), (300, 0, 1.0, 0.5), (40, 1000, 1.0, 1.0),
                                        (129, 50, 0.0, 1.0), (129, 50, 0.0, 2.0)]:
                seed += 1
                c = dict(kind=kind, uplo=uplo, trans=trans, n=n, k=k, alpha=alpha, beta=beta, pad=seed % 3, seed=seed)
                a, c0, want = run_case(ol.oracle_herk(kind), c)
                da = DeviceMatrix.from_numpy(np.array(a), ld=max(a.shape[0], 1) + seed % 3)
                dc = DeviceMatrix.from_numpy(np.array(c0), ld=n + 1)
                herk_dev(uplo, trans, alpha, da, beta, dc)
                got = dc.to_numpy()
                assert herk_ratio(c, a, c0, got, want) < RATIO_LIMIT, c


@pytest.mark.parametrize("kind,uplo,trans", [(3, "L", "N"), (3, "U", "C"), (2, "L", "C")])
def test_large_update_matches_gemm_triangle(kind, uplo, trans):
    n, k = 4096, 2048
    alpha, beta = 0.7, 1.3
    a = DeviceMatrix(n if trans == "N" else k, k if trans == "N" else n, kind).fill_uniform(21)
    c = DeviceMatrix(n, n, kind).fill_uniform(22)
    full = DeviceMatrix(n, n, kind).fill_uniform(22)
    c0 = c.as_strided().clone()
    herk_dev(uplo, trans, alpha, a, beta, c)
    if trans == "N":
        gemm_dev("N", "C", alpha, a, a, beta, full)
    else:
        gemm_dev("C", "N", alpha, a, a, beta, full)
    torch.cuda.synchronize()
    got, ref = c.as_strided(), full.as_strided()
    tri = torch.triu(torch.ones(n, n, dtype=torch.bool, device="cuda")) if uplo == "U" else \
        torch.tril(torch.ones(n, n, dtype=torch.bool, device="cuda"))
    assert torch.equal(got[~tri], c0[~tri])                       # the other triangle is untouched
    assert torch.all(torch.diagonal(got).imag == 0)
    strict = tri & ~torch.eye(n, dtype=torch.bool, device="cuda")
    err = (got[strict] - ref[strict]).abs().max().item()
    assert err <= 20 * k * ol.EPS[kind] * abs(alpha), err
    # the diagonal: real part as in the plain product (whose diagonal also has an imaginary part: beta * Im c0)
    derr = (torch.diagonal(got).real - torch.diagonal(ref).real).abs().max().item()
    assert derr <= 20 * k * ol.EPS[kind] * abs(alpha), derr
