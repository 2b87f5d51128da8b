"""Parity of the HIP path for the two product forms Eigen keeps off the F77 route -- one real
and one complex operand, and a destination with an inner stride -- through the C ABI
(eigamd_gemm_ex_host, eigamd_gemm_mixed_dev): against golden outputs of Eigen's operator*,
against the CPU oracle on fresh inputs, and, at a size the oracle does not reach, the mixed
product against the library's own complex product on a widened operand.
Tolerance: the Netlib test ratio (see test_gemm_gpu.py) below 16."""
import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from eigen_amd import _lib
from eigen_amd.blas import gemm_ex_host
from test_oracle import RATIO_LIMIT
from test_gemm_ex_oracle import golden_ex, ratio_ex, run_case

pytestmark = pytest.mark.gpu


def test_golden_mixed_and_strided_outputs():
    z, meta = golden_ex()
    for idx, c in enumerate(meta["cases"]):
        got, a, b, c0w = run_case(gemm_ex_host, c, meta["seeds"])
        r = ratio_ex(c, got, z[f"res_{idx}"], a, b, c0w)
        assert r < RATIO_LIMIT, (idx, c, r)


@pytest.mark.parametrize("kind", [2, 3])
@pytest.mark.parametrize("real", ["a", "b"])
def test_mixed_host_vs_oracle_all_forms(kind, real):
    seeds = [15, 16, 17]
    scal = [(1.0 + 0j, 1.0 + 0j), (0.7 - 0.9j, 1.3 - 1.1j), (-1.0 + 0j, 0j), (0.5 + 0j, 2.0 + 0j)]
    i = 0
    for (m, n, k) in [(1, 1, 1), (129, 127, 130), (300, 517, 211), (513, 260, 77)]:
        for opa in "NTCJ":
            for opb in "NTCJ":
                i += 1
                alpha, beta = scal[i % 4]
                c = dict(kind=kind, real=real, opa=opa, opb=opb, m=m, n=n, k=k, alpha=[alpha.real, alpha.imag],
                         beta=[beta.real, beta.imag], pad=i % 3, c_inc=1 if i % 3 else 2, stride=1)
                got, a, b, c0w = run_case(gemm_ex_host, c, seeds)
                want, _, _, _ = run_case(ol.oracle_product_ex, c, seeds)
                r = ratio_ex(c, got, want, a, b, c0w)
                assert r < RATIO_LIMIT, (c, r)


@pytest.mark.parametrize("kind", range(4))
def test_strided_destination_vs_oracle(kind):
    seeds = [25, 26, 27]
    for i, (m, n, k) in enumerate([(7, 9, 1), (128, 128, 128), (255, 257, 65), (1000, 40, 77)]):
        for opa, opb in [("N", "N"), ("T", "C"), ("C", "T")]:
            alpha, beta = [(1.0, 1.0), (0.7, 1.3), (-1.0, 0.0)][(i + len(opa)) % 3]
            c = dict(kind=kind, real="", opa=opa, opb=opb, m=m, n=n, k=k, alpha=[alpha, 0.0], beta=[beta, 0.0], pad=i % 2,
                     c_inc=2 + i % 3, stride=1)
            got, a, b, c0w = run_case(gemm_ex_host, c, seeds)
            want, _, _, _ = run_case(ol.oracle_product_ex, c, seeds)
            assert ratio_ex(c, got, want, a, b, c0w) < RATIO_LIMIT, c


def test_ex_argument_errors():
    a = np.zeros((4, 4), dtype=np.float64, order="F")
    z = np.zeros((4, 4), dtype=np.complex128, order="F")
    with pytest.raises(_lib.EigenAmdError):   # two real operands with a complex result
        gemm_ex_host("N", "N", 4, 4, 4, 1.0, a, 4, a, 4, 1.0, z, 1, 4)
    with pytest.raises(_lib.EigenAmdError):   # columns of the strided C overlap
        gemm_ex_host("N", "N", 4, 4, 4, 1.0, a, 4, a, 4, 1.0, a, 2, 4)
    with pytest.raises(_lib.EigenAmdError):   # inner stride 0
        gemm_ex_host("N", "N", 4, 4, 4, 1.0, a, 4, a, 4, 1.0, a, 0, 4)


@pytest.mark.parametrize("kind", [2, 3])
def test_mixed_device_entry_matches_widened_complex_product(kind):
    """4096 x 2048 x 3072 on device-resident operands: the complex x real product (the real-view
    launch for plain A and real alpha, the promoting launch otherwise) against the library's own
    complex kernel fed a widened copy of the real operand."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev, gemm_mixed_dev
    m, n, k = 4096, 2048, 3072
    az = DeviceMatrix(m, k, kind).fill_uniform(1)
    br = DeviceMatrix(k, n, kind - 2).fill_uniform(2)
    bz = DeviceMatrix(k, n, kind)
    bz.as_strided().copy_(br.as_strided())
    ar = DeviceMatrix(m, k, kind - 2).fill_uniform(4)
    arz = DeviceMatrix(m, k, kind)
    arz.as_strided().copy_(ar.as_strided())
    bzz = DeviceMatrix(k, n, kind).fill_uniform(5)
    eps = ol.EPS[kind]
    for (x, y, xw, yw, opa, alpha, beta) in [(az, br, az, bz, "N", -1.0, 1.0),           # real-view launch
                                            (az, br, az, bz, "N", 0.7 - 0.9j, 1.0),     # complex alpha: promoted
                                            (az, br, az, bz, "J", 1.0, 0.5),            # conj(A): promoted
                                            (ar, bzz, arz, bzz, "N", 1.0, 1.0)]:        # real x complex
        c1 = DeviceMatrix(m, n, kind).fill_uniform(3)
        c2 = DeviceMatrix(m, n, kind).fill_uniform(3)
        gemm_mixed_dev(opa, "N", alpha, x, y, beta, c1)
        gemm_dev(opa, "N", alpha, xw, yw, beta, c2)
        torch.cuda.synchronize()
        d = (c1.as_strided() - c2.as_strided()).abs().max().item()
        # |a||b| summed over k is below 2k; the two launches differ in summation order only
        assert d <= 16 * eps * 2 * k * abs(alpha), (opa, alpha, d)
        assert d == d
