"""GPU parity tests of ?symm_, ?hemm_, ?trmm_ and their device-pointer forms through the C
ABI: the restated Netlib ?CHK2 / TRMM rows of ?CHK3 with the reference's parameter files, the
golden results of Eigen's own entry points, the error exits, and the device entries against
the CPU oracle at sizes beyond one GEMM tile.  Tolerance: Netlib test ratio < 16."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from eigen_amd import _lib
from eigen_amd.device import DeviceMatrix, symm_dev, trmm_dev
from test_level3x_oracle import RATIO_LIMIT, abs1_of, check_against_golden, oracle_lookup

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")


def product_lookup(name):
    return getattr(_lib.lib, name)


@pytest.mark.parametrize("kind", range(4))
def test_netlib_known_answers_gpu(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    sfx = ol.SUFFIX[kind]
    rep = ol.run_netlib_chk2(kind, product_lookup(sfx + "symm_"), False, p)
    assert rep.ncalls == 1296 and not rep.fatal and rep.nfail_same == 0 and rep.errmax < p["thresh"] == RATIO_LIMIT
    if kind >= 2:
        rep = ol.run_netlib_chk2(kind, product_lookup(sfx + "hemm_"), True, p)
        assert rep.ncalls == 1296 and not rep.fatal and rep.nfail_same == 0 and rep.errmax < p["thresh"]
    rep = ol.run_netlib_chk3_trmm(kind, product_lookup(sfx + "trmm_"), p)
    assert rep.ncalls == 2592 and not rep.fatal and rep.nfail_same == 0 and rep.errmax < p["thresh"]


def test_golden_reference_results():
    check_against_golden(product_lookup)


def test_error_exits():
    """One call per argument check (blas/level3_impl.h:256-266, :213-223): the code reaches
    xerbla_ with the routine's name and nothing is written."""
    x = np.zeros((2, 2), order="F")
    one = np.array([1.0])
    ci = lambda v: ctypes.byref(ctypes.c_int(v))
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    for info, (sd, ul, m, n, lda, ldb, ldc) in {1: (b"/", b"U", 0, 0, 1, 1, 1), 2: (b"L", b"/", 0, 0, 1, 1, 1),
                                                 3: (b"L", b"U", -1, 0, 1, 1, 1), 4: (b"L", b"U", 0, -1, 1, 1, 1),
                                                 7: (b"L", b"U", 2, 0, 1, 2, 2), 9: (b"L", b"U", 2, 0, 2, 1, 2),
                                                 12: (b"L", b"U", 2, 0, 2, 2, 1)}.items():
        ol.reset_xerbla()
        _lib.lib.dsymm_(sd, ul, ci(m), ci(n), vp(one), vp(x), ci(lda), vp(x), ci(ldb), vp(one), vp(x), ci(ldc))
        assert ol.last_xerbla() == (info, "DSYMM ")
    for info, (sd, ul, tr, dg, m, n, lda, ldb) in {1: (b"/", b"U", b"N", b"N", 0, 0, 1, 1), 2: (b"L", b"/", b"N", b"N", 0, 0, 1, 1),
                                                    3: (b"L", b"U", b"/", b"N", 0, 0, 1, 1), 4: (b"L", b"U", b"N", b"/", 0, 0, 1, 1),
                                                    5: (b"L", b"U", b"N", b"N", -1, 0, 1, 1), 6: (b"L", b"U", b"N", b"N", 0, -1, 1, 1),
                                                    9: (b"L", b"U", b"N", b"N", 2, 0, 1, 2), 11: (b"L", b"U", b"N", b"N", 2, 0, 2, 1)}.items():
        ol.reset_xerbla()
        _lib.lib.dtrmm_(sd, ul, tr, dg, ci(m), ci(n), vp(one), vp(x), ci(lda), vp(x), ci(ldb))
        assert ol.last_xerbla() == (info, "DTRMM ")
    ol.reset_xerbla()


@pytest.mark.parametrize("kind", range(4))
def test_device_entries_vs_oracle(kind):
    """520-term sums here (9 in the Netlib cases) and two computed results compared with each
    other: the limit on the componentwise ratio is 4 x 16."""
    limit = 4 * RATIO_LIMIT
    abs1 = abs1_of(kind)
    alpha, beta = ((0.7 - 0.9j), (1.3 - 1.1j)) if kind >= 2 else (0.7, 1.3)
    seed = 500
    for side in "LR":
        for uplo in "UL":
            m, n = (520, 300) if side == "L" else (300, 520)
            for herm in ([False, True] if kind >= 2 else [False]):
                seed += 1
                a, b, c = ol.symm_operands(kind, side, uplo, herm, m, n, seed)
                c0, want = np.array(c), np.asfortranarray(np.array(c))
                name = ol.SUFFIX[kind] + ("hemm_" if herm else "symm_")
                ol.call_f77_symm(oracle_lookup(name), side, uplo, m, n, alpha, np.asfortranarray(np.array(a)), a.shape[0],
                                 np.asfortranarray(np.array(b)), m, beta, want, m)
                da, db = DeviceMatrix.from_numpy(np.array(a)), DeviceMatrix.from_numpy(np.array(b), ld=m + 3)
                dc = DeviceMatrix.from_numpy(np.array(c), ld=m + 1)
                symm_dev(side, uplo, alpha, da, db, beta, dc, hermitian=herm)
                s = abs1(ol.dense_symmetric(np.array(a), uplo, herm)).astype(np.float64)
                g = abs(alpha) * (s @ abs1(np.array(b)) if side == "L" else abs1(np.array(b)) @ s) + abs(beta) * abs1(c0)
                r = (abs1(dc.to_numpy().astype(np.complex128) - want) / (ol.EPS[kind] * g)).max()
                assert r < limit, (side, uplo, herm, r)
            for trans in "NTC":
                for diag in "NU":
                    seed += 1
                    a, b = ol.trsm_operands(kind, side, uplo, diag, m, n, seed)
                    want = np.asfortranarray(np.array(b))
                    ol.call_f77_trsm(oracle_lookup(ol.SUFFIX[kind] + "trmm_"), side, uplo, trans, diag, m, n, alpha,
                                     np.asfortranarray(np.array(a)), a.shape[0], want, m)
                    da, db = DeviceMatrix.from_numpy(np.array(a), ld=a.shape[0] + 1), DeviceMatrix.from_numpy(np.array(b))
                    trmm_dev(side, uplo, trans, diag, alpha, da, db)
                    t = abs1(ol.dense_op_triangle(np.array(a), uplo, trans, diag)).astype(np.float64)
                    g = abs(alpha) * (t @ abs1(np.array(b)) if side == "L" else abs1(np.array(b)) @ t)
                    g = np.where(g == 0, 1.0, g)
                    r = (abs1(db.to_numpy().astype(np.complex128) - want) / (ol.EPS[kind] * g)).max()
                    assert r < limit, (side, uplo, trans, diag, r)
                    assert np.array_equal(da.to_numpy(), np.array(a))   # rogue triangle untouched
