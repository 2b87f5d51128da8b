"""The LU code paths that only large or forced configurations reach, each against a checker:

* the configured size (PartialPivLU 32768 x 32768 double): outer steps 512 / 1024 columns wide,
  look-ahead on two streams -- properties every partial-pivoting LU has plus the reference's
  own acceptance criterion (test/lu.cpp:168) on sampled rows of P*A - L*U;
* the same wide outer steps forced at sizes the CPU oracle finishes in seconds
  (EIGAMD_LU_WIDE), against oracle/lu_oracle.c (Eigen/src/LU/PartialPivLU.h:429-498);
* the trailing update dealt over two streams (EIGAMD_LU_SPLIT_MIN) forced at small sizes: against the oracle and,
  bit for bit, against the one-stream form;
* the per-column kernel chain (EIGAMD_LU_FUSED=0, EIGAMD_LU_SINGLE=0), including a column of
  NaNs, and the second attempt of the host entry after a failed one-launch base case.
"""
import os

import numpy as np
import pytest

import eigen_amd
from eigen_amd import lu as ealu
from lu_checks import check_against_oracle, check_structure, perm_of, residual
from test_lu_oracle import oracle_lu

pytestmark = pytest.mark.gpu


class env:
    """Environment variables the library reads at every factorisation."""

    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def sampled_lu_residual(n, a_dev, a0_dev, transp, nblocks=6, block=128, seed=0):
    """max over sampled blocks of rows R of ||(P A)[R, :] - L[R, :] U||_F / ||(P A)[R, :]||_F."""
    import torch
    lu = a_dev.as_strided()
    perm = torch.as_tensor(perm_of(transp, n), device="cuda")
    u = torch.triu(lu)
    rng = np.random.default_rng(seed)
    starts = sorted({0, n - block, *[int(x) for x in rng.integers(0, n - block, nblocks)]})
    worst = 0.0
    cols = torch.arange(n, device="cuda")
    for r0 in starts:
        rows = torch.arange(r0, r0 + block, device="cuda")
        lr = lu[rows]                                              # block x n
        lr = torch.where(cols[None, :] < rows[:, None], lr, torch.zeros_like(lr))
        lr[torch.arange(block, device="cuda"), rows] = 1.0
        pa = a0_dev.as_strided()[perm[rows]]
        worst = max(worst, float(torch.linalg.norm(pa - lr @ u) / torch.linalg.norm(pa)))
    return worst


def test_lu_configured_size_32768():
    import torch
    from eigen_amd.device import DeviceMatrix
    n = 32768
    a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(77)
    transp, nb, fz = ealu.getrf_dev(a)
    assert fz == -1
    t = transp.cpu().numpy()
    assert (t >= np.arange(n)).all() and (t < n).all()
    assert nb == int((t != np.arange(n)).sum())
    lu = a.as_strided()
    lmax = 0.0
    for c0 in range(0, n, 4096):                                   # |l_ij| <= 1, in column chunks
        lmax = max(lmax, float(torch.tril(lu[:, c0:c0 + 4096], -1 - c0).abs().max()))
    assert lmax <= 1.0, lmax
    a0 = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(77)
    res = sampled_lu_residual(n, a, a0, t)
    assert res <= 8 * np.finfo(np.float64).eps * n, res


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("wide,n", [(512, 2048), (1024, 2048), (768, 2304), (1024, 2560)])
def test_forced_wide_outer_steps_vs_oracle(dt, wide, n):
    a = eigen_amd.fill_uniform_host(dt, n, n, 400 + n + wide)
    a1, a2 = a.copy(order="F"), a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1)
    with env(EIGAMD_LU_WIDE=wide):
        t2, nb2, r2 = ealu.blocked_lu(a2)
    assert r1 == r2 == -1
    first_diff = check_against_oracle(a, a1, t1, a2, t2, exact_pivots=dt == np.float64)
    if first_diff is None:
        assert nb1 == nb2


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("split_min,wide,n,slice_cols", [(256, 0, 3072, 0), (512, 512, 3072, 0), (256, 256, 2560, 0), (1024, 1024, 4096, 0),
                                                         (256, 0, 3072, 256), (0, 512, 4096, 512)])
def test_trailing_update_dealt_over_two_streams_vs_oracle(dt, split_min, wide, n, slice_cols):
    """The bulk of a step's trailing update on two streams (EIGAMD_LU_SPLIT_MIN columns and more remaining; 8192 by
    default, so only matrices from about 9000 up take it): forced at sizes the oracle finishes in seconds, with the
    ramp of the first steps, the columns dealt anew as the left part shrinks, and the return to one stream; with
    slice_cols, the panel stream also updates the last columns of the trailing matrix itself (EIGAMD_LU_SLICE)."""
    a = eigen_amd.fill_uniform_host(dt, n, n, 900 + n + wide)
    a1, a2 = a.copy(order="F"), a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1)
    knobs = dict(EIGAMD_LU_SPLIT_MIN=split_min, EIGAMD_LU_SLICE=slice_cols, EIGAMD_LU_SLICE_MIN=1024)
    if wide:
        knobs["EIGAMD_LU_WIDE"] = wide
    with env(**knobs):
        t2, nb2, r2 = ealu.blocked_lu(a2)
    assert r1 == r2 == -1
    first_diff = check_against_oracle(a, a1, t1, a2, t2, exact_pivots=dt == np.float64)
    if first_diff is None:
        assert nb1 == nb2


def test_split_and_unsplit_updates_give_the_same_bits():
    import torch
    from eigen_amd.device import DeviceMatrix
    n = 6144
    out = []
    # (last: with the panel stream updating the last 512 columns itself while 2048 remain)
    for split_min, slice_cols in ((0, 0), (256, 0), (2048, 0), (256, 512), (0, 512)):
        with env(EIGAMD_LU_SPLIT_MIN=split_min, EIGAMD_LU_SLICE=slice_cols, EIGAMD_LU_SLICE_MIN=2048):
            a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(31)
            transp, nb, fz = ealu.getrf_dev(a)
            torch.cuda.synchronize()
            out.append((a.as_strided().clone(), transp.clone(), nb, fz))
    for lu, t, nb, fz in out[1:]:
        assert torch.equal(t, out[0][1]) and nb == out[0][2] and fz == out[0][3]
        assert torch.equal(lu, out[0][0])


def test_wide_steps_equal_narrow_steps_on_device():
    """The same matrix through outer steps of 256, 512 and 1024 columns: identical transpositions,
    factors equal to rounding (a k = 1024 product sums what four k = 256 products sum)."""
    from eigen_amd.device import DeviceMatrix
    n = 6144
    out = []
    for wide in (256, 512, 1024):
        a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(91)
        with env(EIGAMD_LU_WIDE=wide):
            transp, nb, fz = ealu.getrf_dev(a)
        out.append((transp.cpu().numpy(), a.to_numpy(), nb, fz))
    for t, lu, nb, fz in out[1:]:
        assert np.array_equal(t, out[0][0]) and nb == out[0][2] and fz == out[0][3] == -1
        assert np.abs(lu - out[0][1]).max() <= 64 * np.finfo(np.float64).eps * n * np.abs(out[0][1]).max()


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_per_column_kernel_chain_vs_oracle(dt):
    for (m, n) in [(300, 300), (1500, 1100), (700, 40)]:
        a = eigen_amd.fill_uniform_host(dt, m, n, 9 + m)
        a1, a2 = a.copy(order="F"), a.copy(order="F")
        t1, nb1, r1 = oracle_lu(a1)
        with env(EIGAMD_LU_FUSED=0, EIGAMD_LU_SINGLE=0):
            t2, nb2, r2 = ealu.blocked_lu(a2)
        assert r1 == r2 == -1
        check_against_oracle(a, a1, t1, a2, t2, exact_pivots=dt == np.float64)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_both_forms_of_the_one_launch_base_case_give_the_same_bits(dt):
    """The one-launch base case either fetches the winner's row once the winner is known or has three of its wavefronts read
    every candidate's row beside the poll (up to 64 workgroups; EIGAMD_LU_WIDE_POLL=0 switches that off): same pivots as
    the oracle and the same factors bit for bit, with one-workgroup base cases (the default below 2048 rows) and without."""
    for (m, n) in [(2304, 2304), (5000, 300), (700, 700)]:
        a = eigen_amd.fill_uniform_host(dt, m, n, 21 + m)
        a0, a1, a2, a3 = (a.copy(order="F") for _ in range(4))
        t0, nb0, r0 = oracle_lu(a0)
        t1, nb1, r1 = ealu.blocked_lu(a1)
        with env(EIGAMD_LU_WIDE_POLL=0):
            t2, nb2, r2 = ealu.blocked_lu(a2)
        with env(EIGAMD_LU_SINGLE=0):
            t3, nb3, r3 = ealu.blocked_lu(a3)
        assert r0 == r1 == r2 == r3 == -1
        check_against_oracle(a, a0, t0, a1, t1, exact_pivots=dt == np.float64)
        assert np.array_equal(t1, t2) and np.array_equal(t1, t3) and nb1 == nb2 == nb3
        assert np.array_equal(a1, a2) and np.array_equal(a1, a3)


@pytest.mark.parametrize("knobs", [dict(EIGAMD_LU_FUSED=0, EIGAMD_LU_SINGLE=0), dict(EIGAMD_LU_SINGLE=0), dict(EIGAMD_LU_SINGLE=0, EIGAMD_LU_WIDE_POLL=0), dict()])
def test_column_of_nans_takes_the_diagonal_entry(knobs):
    """No comparable candidate in a column: every base case keeps the diagonal entry as the pivot,
    exchanges nothing and stays inside the matrix (transposition = the column's own index)."""
    n = 600
    a = eigen_amd.fill_uniform_host(np.float64, n, n, 3)
    a[:, 5] = np.nan
    with env(**knobs):
        t, nb, fz = ealu.blocked_lu(a)
    assert t[5] == 5 and (t >= np.arange(n)).all() and (t < n).all()


def test_host_entry_runs_again_without_the_fused_base_case():
    """A failed one-launch base case (its workgroups not resident together) leaves the device copy
    partly overwritten; the host entries still hold the caller's matrix and factorise it again
    through the per-column kernels.  The failure is injected after the first attempt."""
    m, n = 1500, 1100
    a = eigen_amd.fill_uniform_host(np.float64, m, n, 5 + m)
    a1, a2 = a.copy(order="F"), a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1)
    with env(EIGAMD_LU_INJECT_FUSED_FAILURE=1, EIGAMD_LU_SINGLE=0):
        t2, nb2, r2 = ealu.blocked_lu(a2)
    check_against_oracle(a, a1, t1, a2, t2, exact_pivots=True)
    assert nb1 == nb2 and r1 == r2 == -1


def test_device_entry_reports_not_resident_and_the_shared_form_finishes_the_job():
    """The device entry's contract for a shared GPU: EIGAMD_ERR_NOT_RESIDENT (code -4, a name of its own), the
    matrix partly overwritten; the caller restores it and calls eigamd_getrf_dev_shared, which gives the oracle's
    transpositions and factors and cannot return that code (the injected failure does not reach it)."""
    import torch
    from eigen_amd.device import DeviceMatrix
    m, n = 1500, 1100
    a = eigen_amd.fill_uniform_host(np.float64, m, n, 5 + m)
    a1 = a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1)
    d = DeviceMatrix.from_numpy(a)
    with env(EIGAMD_LU_INJECT_FUSED_FAILURE=1, EIGAMD_LU_SINGLE=0):
        with pytest.raises(ealu.NotResident):
            ealu.getrf_dev(d)
        d = DeviceMatrix.from_numpy(a)   # the caller's copy
        transp, nb2, r2 = ealu.getrf_dev(d, shared=True)
    check_against_oracle(a, a1, t1, d.to_numpy(), transp.cpu().numpy(), exact_pivots=True)
    assert nb1 == nb2 and r1 == r2 == -1


def test_host_entry_beside_a_long_product_on_another_stream():
    """The device shared for real: a queue of large products on another stream fills every compute unit's
    registers while the factorisation runs.  Whether the one-launch base case still gets its workgroups
    resident together or gives up and the host entry runs again through the per-column kernels, the
    result must be the oracle's."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    n = 8192
    pa = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(1)
    pb = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(2)
    pc = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(3)
    m = 2304
    a = eigen_amd.fill_uniform_host(np.float64, m, m, 77)
    a1, a2 = a.copy(order="F"), a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(12):  # about 0.2 s of products
        gemm_dev("N", "N", 1.0, pa, pb, 0.5, pc, stream=side)
    with env(EIGAMD_LU_SINGLE=0):
        t2, nb2, r2 = ealu.blocked_lu(a2)
    busy = not side.query()
    torch.cuda.synchronize()
    print("products still running when the factorisation returned:", busy)
    check_against_oracle(a, a1, t1, a2, t2, exact_pivots=True)
    assert nb1 == nb2 and r1 == r2 == -1


def steep_lower_times_upper(n, rng, dt):
    """A = L U with L unit lower, -0.9 everywhere below the diagonal (inv(L) has entries up to
    0.9 * 1.9^(n-2)), and U a random upper triangle with a dominant diagonal: the diagonal entry
    is the strict maximum of every column, so partial pivoting exchanges nothing at first and
    the unit triangles the factorisation solves with are (perturbations of) L's blocks."""
    l = np.tril(-0.9 * np.ones((n, n)), -1) + np.eye(n)
    u = np.triu(rng.uniform(-1, 1, (n, n)), 1) + np.diag(4.0 + rng.uniform(0, 1, n))
    return np.asfortranarray(l @ u).astype(dt)


@pytest.mark.parametrize("dt,n", [(np.float64, 600), (np.float64, 2304), (np.float32, 600)])
def test_ill_conditioned_unit_triangle_inside_lu(dt, n):
    """The factors of such a matrix are not determined to working precision (the oracle and the
    device return different, equally valid ones), the residual is: test/lu.cpp:168's criterion
    for both, and the same criterion failing by orders of magnitude once every solve multiplies
    by the explicit inverse (guard switched off)."""
    rng = np.random.default_rng(n)
    a = steep_lower_times_upper(n, rng, dt)
    eps = np.finfo(dt).eps
    a1, a2, a3 = a.copy(order="F"), a.copy(order="F"), a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1)
    t2, nb2, r2 = ealu.blocked_lu(a2)
    res_or, res_dev = residual(a, a1, t1), residual(a, a2, t2)
    with env(EIGAMD_LU_INVERSE_LIMIT="inf"):
        t3, nb3, r3 = ealu.blocked_lu(a3)
    res_inv = residual(a, np.nan_to_num(a3, nan=0.0, posinf=0.0, neginf=0.0), t3)
    print(f"{dt.__name__} {n}: residual / (eps n): oracle {res_or / eps / n:.3g}, device {res_dev / eps / n:.3g}, "
          f"device with explicit inverses only {res_inv / eps / n:.3g}")
    check_structure(a2, t2, n, n)
    assert res_dev <= 8 * eps * n and res_or <= 8 * eps * n
    assert res_inv > 1e3 * res_dev


@pytest.mark.parametrize("dt,n", [(np.float64, 64), (np.float64, 200), (np.float64, 512), (np.float32, 100)])
def test_growth_matrix(dt, n):
    """Wilkinson's example: 1 on the diagonal and in the last column, -1 below the diagonal.  No
    exchanges, L = the lower part, the last column of U doubles from row to row."""
    a = np.asfortranarray(np.tril(-np.ones((n, n)), -1) + np.eye(n)).astype(dt)
    a[:, -1] = 1.0
    a1, a2 = a.copy(order="F"), a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1)
    t2, nb2, r2 = ealu.blocked_lu(a2)
    assert np.array_equal(t1, t2) and nb1 == nb2 == 0
    assert np.array_equal(a1, a2)
    assert a2[-1, -1] == dt(2.0) ** (n - 1)


def test_substitution_path_on_ordinary_matrices():
    """Every solve of the blocked LU through the substitution kernel (limit 0) against the oracle."""
    for dt in (np.float64, np.float32):
        for (m, n) in [(700, 700), (2100, 2100)]:
            a = eigen_amd.fill_uniform_host(dt, m, n, 5 + m)
            a1, a2 = a.copy(order="F"), a.copy(order="F")
            t1, nb1, r1 = oracle_lu(a1)
            with env(EIGAMD_LU_INVERSE_LIMIT=0):
                t2, nb2, r2 = ealu.blocked_lu(a2)
            check_against_oracle(a, a1, t1, a2, t2, exact_pivots=dt == np.float64)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_rowmajor_storage_gives_the_same_factorisation(dt):
    """partial_lu_impl<Scalar, RowMajor> (Eigen/src/LU/PartialPivLU.h:333-341): the matrix stored row by
    row, with a row stride larger than its width, factorises to the same factors and transpositions."""
    for (m, n) in [(300, 300), (700, 500), (200, 450)]:
        a = eigen_amd.fill_uniform_host(dt, m, n, 31 + m)
        col = a.copy(order="F")
        t1, nb1, r1 = ealu.blocked_lu(col)
        buf = np.full((m, n + 5), -7.0, dtype=dt)           # C order, row stride n + 5
        buf[:, :n] = a
        t2, nb2, r2 = ealu.blocked_lu_rowmajor(buf[:, :n])
        assert np.array_equal(t1, t2) and nb1 == nb2 and r1 == r2
        assert np.array_equal(buf[:, :n], col) and (buf[:, n:] == -7.0).all()


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("m,n", [(3000, 3000), (1000, 2000), (2000, 700)])
def test_device_entry_with_blocks_of_sixteen_on_a_wide_matrix(dt, m, n):
    """max_block_size = 16 hands the whole matrix to the panel driver: hundreds of 32-column blocks right of a
    16-column block, far more than a panel of the look-ahead ever has (the one-launch form of the row
    operations and the rank update keeps a counter per such block and only takes what its counters cover)."""
    import torch
    from eigen_amd.device import DeviceMatrix
    a = eigen_amd.fill_uniform_host(dt, m, n, 1300 + m + n)
    a1 = a.copy(order="F")
    t1, nb1, r1 = oracle_lu(a1, max_block=16)
    d = DeviceMatrix(m, n, eigen_amd.KIND_OF_DTYPE[np.dtype(dt)])
    d.as_strided().copy_(torch.from_numpy(np.ascontiguousarray(a)))
    transp, nb2, r2 = ealu.getrf_dev(d, max_block_size=16)
    a2 = np.asfortranarray(d.as_strided().cpu().numpy())
    t2 = transp.cpu().numpy()
    assert r1 == r2 == -1
    first_diff = check_against_oracle(a, a1, t1, a2, t2, exact_pivots=dt == np.float64)
    if first_diff is None:
        assert nb1 == nb2


def test_calls_that_do_not_wait_share_the_scratch_in_order():
    """eigamd_getrf_dev_async returns with its kernels queued; a second factorisation from another stream uses the
    same per-device scratch (counters, plans, inverses, exchange area) and must run behind the first."""
    import torch
    from eigen_amd.device import DeviceMatrix
    n = 2304
    mats, want = [], []
    for i in range(3):
        a = eigen_amd.fill_uniform_host(np.float64, n, n, 1700 + i)
        a1 = a.copy(order="F")
        t1, nb1, r1 = oracle_lu(a1)
        d = DeviceMatrix(n, n, eigen_amd.F64)
        d.as_strided().copy_(torch.from_numpy(np.ascontiguousarray(a)))
        mats.append((a, d))
        want.append((a1, t1, nb1))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in mats]
    outs = []
    for (a, d), s in zip(mats, streams):
        transp = torch.zeros(n, dtype=torch.int32, device="cuda")
        info = torch.zeros(3, dtype=torch.int32, device="cuda")
        ealu.getrf_dev_async(d, transp, info, stream=s)
        outs.append((transp, info))
    torch.cuda.synchronize()
    for (a, d), (a1, t1, nb1), (transp, info) in zip(mats, want, outs):
        fz, nb2, bad = (int(v) for v in info.cpu())
        assert (fz, bad) == (-1, 0)
        a2 = np.asfortranarray(d.as_strided().cpu().numpy())
        assert check_against_oracle(a, a1, t1, a2, transp.cpu().numpy(), exact_pivots=True) is None
        assert nb1 == nb2


@pytest.mark.parametrize("kind_name,n,runs", [("f64", 4096, 6), ("c64", 2304, 4), ("f32", 3000, 4)])
def test_every_run_gives_the_same_bits(kind_name, n, runs):
    """The factorisation has workgroups that ride in other kernels' launches, arrival counters, memory-side adds and four
    streams; none of it may make the result depend on timing: transpositions and factors of repeated runs are equal bit for bit."""
    import torch
    from eigen_amd.device import DeviceMatrix
    kind = {"f32": eigen_amd.F32, "f64": eigen_amd.F64, "c64": eigen_amd.C64}[kind_name]
    a = DeviceMatrix(n, n, kind)
    first = None
    for _ in range(runs):
        a.fill_uniform(77)
        transp, nb, fz = ealu.getrf_dev(a)
        torch.cuda.synchronize()
        lu = a.as_strided()
        cur = (transp.clone(), (torch.view_as_real(lu) if kind >= 2 else lu).clone(), nb, fz)
        if first is None:
            first = cur
        else:
            assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1]) and cur[2:] == first[2:]
