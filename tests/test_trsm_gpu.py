"""GPU parity tests of the triangular solve (?trsm_ and eigamd_trsm_dev) through the C ABI:
the restated Netlib ?CHK3 / ?CHKE rows with the reference's parameter files, the golden
solutions of Eigen's ?trsm_, the CPU oracle on every side/uplo/trans/diag form, and a
residual property at sizes the oracle does not reach.

Tolerance: Netlib test ratio |op(A) X - alpha B| / (eps * (|op(A)||X| + |alpha||B|)) < 16;
against reference/oracle solutions max|x - x_ref| <= 64 * eps * na * max|x_ref| (well
conditioned triangles)."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_lib as ol
import eigen_amd
from eigen_amd import _lib
from eigen_amd.device import DeviceMatrix, trsm_dev
from test_lu_config_gpu import env

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
NETLIB = os.path.join(HERE, "golden", "netlib")
RATIO_LIMIT = 16.0


def f77(kind):
    return getattr(_lib.lib, ol.SUFFIX[kind] + "trsm_")


@pytest.mark.parametrize("kind", range(4))
def test_netlib_trsm_known_answers_gpu(kind):
    p = ol.parse_blat3_dat(os.path.join(NETLIB, f"{ol.SUFFIX[kind]}blat3.dat"))
    rep = ol.run_netlib_chk3_trsm(kind, f77(kind), p)
    assert rep.ncalls == 2592
    assert not rep.fatal and rep.nfail_same == 0 and rep.nfail_ratio == 0 and rep.nfail_xerbla == 0
    assert rep.errmax < p["thresh"] == RATIO_LIMIT
    assert ol.run_netlib_chke_trsm(kind, f77(kind)) == 0


def test_golden_reference_solutions():
    z = np.load(os.path.join(HERE, "golden", "trsm_golden.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    for idx, c in enumerate(meta["cases"]):
        a, b = ol.trsm_operands(c["kind"], c["side"], c["uplo"], c["diag"], c["m"], c["n"], c["seed"], c["pad"])
        b0 = np.array(b)
        alpha = complex(*c["alpha"]) if c["kind"] >= 2 else c["alpha"][0]
        na = c["m"] if c["side"] == "L" else c["n"]
        ol.call_f77_trsm(f77(c["kind"]), c["side"], c["uplo"], c["trans"], c["diag"], c["m"], c["n"], alpha, a,
                         na + c["pad"], b, c["m"] + c["pad"])
        x = np.array(b)
        want = z[f"x{idx}"]
        got = x[::c["stride"], ::c["stride"]]
        assert np.abs(got - want).max() <= 64 * ol.EPS[c["kind"]] * na * max(np.abs(want).max(), 1e-30), (idx, c)
        assert ol.trsm_residual_ratio(c["kind"], c["side"], c["uplo"], c["trans"], c["diag"], alpha, a, b0, x) < RATIO_LIMIT


@pytest.mark.parametrize("kind", range(4))
def test_device_entry_vs_oracle_all_forms(kind):
    """Every form on device-resident operands, extents that are not multiples of the 128-wide
    diagonal block, leading dimensions that rule out the vector-load kernels."""
    dt = ol.DTYPES[kind]
    alpha = (0.7 - 0.9j) if kind >= 2 else 0.7
    seed = 40
    for side in "LR":
        for uplo in "UL":
            for trans in "NTC":
                for diag in "NU":
                    seed += 1
                    m, n = (300, 170) if side == "L" else (170, 300)
                    a, b = ol.trsm_operands(kind, side, uplo, diag, m, n, seed, pad=seed % 3)
                    na = a.shape[0]
                    want = np.array(b)
                    wbuf = np.asfortranarray(np.zeros((m + seed % 3, n), dtype=dt))
                    wbuf[:m] = want
                    ol.call_f77_trsm(ol.oracle_trsm(kind), side, uplo, trans, diag, m, n, alpha, a, na + seed % 3, wbuf,
                                     m + seed % 3)
                    da = DeviceMatrix.from_numpy(np.array(a), ld=na + seed % 3)
                    db = DeviceMatrix.from_numpy(np.array(b), ld=m + 1)
                    trsm_dev(side, uplo, trans, diag, alpha, da, db)
                    got = db.to_numpy()
                    ref = wbuf[:m]
                    assert np.abs(got - ref).max() <= 64 * ol.EPS[kind] * na * np.abs(ref).max(), (side, uplo, trans, diag)
                    # the triangle that must not be read still holds the rogue value
                    assert np.array_equal(da.to_numpy(), np.array(a))


@pytest.mark.parametrize("kind", range(4))
def test_leaves_of_two_blocks_all_forms_vs_leaves_of_one(kind):
    """The leaves of the recursion are pairs of diagonal blocks whose inverse is put together from the blocks' inverses
    (EIGAMD_TRSM_PAIRS=0: single blocks).  Every form over several pairs and a ragged last block (five blocks of 256 for
    the real kinds, nine of 128 for the complex ones): the Netlib test ratio against the operands, and the two forms
    against each other to the tolerance of the oracle comparisons."""
    alpha = (0.7 - 0.9j) if kind >= 2 else 0.7
    seed = 240
    for side in "LR":
        for uplo in "UL":
            for trans in "NTC":
                for diag in "NU":
                    seed += 1
                    m, n = (1100, 130) if side == "L" else (130, 1100)
                    a, b = ol.trsm_operands(kind, side, uplo, diag, m, n, seed, pad=seed % 3)
                    na = a.shape[0]
                    out = []
                    for pairs in (1, 0):
                        da = DeviceMatrix.from_numpy(np.array(a), ld=na + seed % 3)
                        db = DeviceMatrix.from_numpy(np.array(b), ld=m + 1)
                        with env(EIGAMD_TRSM_PAIRS=pairs):
                            trsm_dev(side, uplo, trans, diag, alpha, da, db)
                        out.append(db.to_numpy())
                    assert ol.trsm_residual_ratio(kind, side, uplo, trans, diag, alpha, a, b, out[0]) < RATIO_LIMIT, (side, uplo, trans, diag)
                    assert ol.trsm_residual_ratio(kind, side, uplo, trans, diag, alpha, a, b, out[1]) < RATIO_LIMIT, (side, uplo, trans, diag)
                    assert np.abs(out[0] - out[1]).max() <= 64 * ol.EPS[kind] * na * np.abs(out[1]).max(), (side, uplo, trans, diag)


@pytest.mark.parametrize("kind,side,uplo,trans,diag", [(1, "L", "L", "N", "U"), (1, "R", "U", "T", "N"),
                                                       (0, "L", "U", "N", "N"), (3, "L", "L", "C", "N"),
                                                       (2, "R", "L", "N", "N")])
def test_large_solve_residual(kind, side, uplo, trans, diag):
    """4096-wide triangle, 2048 right-hand sides: sampled rows of op(A) X against alpha B,
    accumulated on the host in double precision.  The sums have 4096 terms, so the limit on
    the componentwise ratio is sqrt(na) = 64 rather than the 16 of the 9-term Netlib cases."""
    na, nrhs = 4096, 2048
    m, n = (na, nrhs) if side == "L" else (nrhs, na)
    alpha = (0.7 - 0.9j) if kind >= 2 else -1.3
    a, b = ol.trsm_operands(kind, side, uplo, diag, m, n, 77, rogue=False)
    da, db = DeviceMatrix.from_numpy(np.array(a)), DeviceMatrix.from_numpy(np.array(b))
    trsm_dev(side, uplo, trans, diag, alpha, da, db)
    torch.cuda.synchronize()
    wide = np.complex128 if kind >= 2 else np.float64
    x = db.to_numpy().astype(wide)
    abs1 = (lambda z: np.abs(z.real) + np.abs(z.imag)) if kind >= 2 else np.abs
    topd = ol.dense_op_triangle(a, uplo, trans, diag).astype(wide)
    rows = np.arange(0, m, 97)
    if side == "L":
        res = topd[rows] @ x - alpha * b[rows].astype(wide)
        g = abs1(topd[rows]) @ abs1(x) + abs(alpha) * abs1(b[rows])
    else:
        res = x[rows] @ topd - alpha * b[rows].astype(wide)
        g = abs1(x[rows]) @ abs1(topd) + abs(alpha) * abs1(b[rows])
    ratio = (abs1(res) / (ol.EPS[kind] * g)).max()
    assert ratio < np.sqrt(na), ratio


@pytest.mark.parametrize("kind", range(4))
def test_host_entry_64bit_vs_oracle(kind):
    """eigamd_trsm_host (what the header seam's triangular_solve_matrix calls): host operands with padded
    leading dimensions, a few forms on both sides, against the oracle; padding rows of B untouched."""
    import ctypes
    dt = ol.DTYPES[kind]
    alpha = np.array([(0.7 - 0.9j) if kind >= 2 else 0.7], dtype=dt)
    SIDE, UPLO, OP, DIAG = {"L": 0, "R": 1}, {"U": 0, "L": 1}, {"N": 0, "T": 1, "C": 2}, {"N": 0, "U": 1}
    vp = lambda x: ctypes.c_void_p(x.ctypes.data)
    for seed, (side, uplo, trans, diag, m, n) in enumerate([("L", "L", "N", "U", 300, 170), ("L", "U", "C", "N", 513, 40),
                                                            ("R", "U", "N", "N", 170, 300), ("R", "L", "T", "U", 77, 600)]):
        a, b = ol.trsm_operands(kind, side, uplo, diag, m, n, 900 + seed, pad=2)
        na = a.shape[0]
        bbuf = np.asfortranarray(np.full((m + 2, n), -1.0e10, dtype=dt))
        bbuf[:m] = b
        want = bbuf.copy(order="F")
        ol.call_f77_trsm(ol.oracle_trsm(kind), side, uplo, trans, diag, m, n, alpha[0], a, na + 2, want, m + 2)
        rc = _lib.lib.eigamd_trsm_host(kind, SIDE[side], UPLO[uplo], OP[trans], DIAG[diag], m, n, vp(alpha), vp(a), na + 2,
                                       vp(bbuf), m + 2)
        _lib.check(rc, "eigamd_trsm_host")
        assert np.array_equal(bbuf[m:], want[m:]), "padding rows of B were written"
        assert np.abs(bbuf[:m] - want[:m]).max() <= 64 * ol.EPS[kind] * na * np.abs(want[:m]).max(), (side, uplo, trans, diag)
