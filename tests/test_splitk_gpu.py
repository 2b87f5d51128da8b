"""Products of few tiles, whose reduction the launcher cuts into pieces run by different workgroups
(partial products into library-owned memory, then one pass that adds them up in a fixed order):
parity with the CPU oracle at the Netlib tolerance (test ratio < 16, see test_gemm_gpu.py), on
whole-tile and ragged extents, every operand form, real and complex alpha / beta; the same bits on
every run; beta == 0 never reads C; products queued on several streams at once do not share
partial-product memory."""
import numpy as np
import pytest

import oracle_lib as ol
import eigen_amd
from test_gemm_gpu import ratio
from test_oracle import RATIO_LIMIT, opmat

pytestmark = pytest.mark.gpu

# (m, n, k): one to 48 tiles, k from 8 k-tiles up
SHAPES = [(512, 512, 1024), (256, 256, 4096), (256, 128, 2000), (1024, 768, 512), (300, 200, 1500), (64, 64, 8192),
          (1000, 1000, 1000), (128, 16, 3001)]


@pytest.mark.parametrize("kind", range(4))
@pytest.mark.parametrize("ops", ["NN", "TN", "NT", "TT", "CN", "NC", "CC"])
def test_split_reduction_vs_oracle(kind, ops):
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
        c = DeviceMatrix(m, n, kind, ld=m + pad).fill_uniform(37 + si)
        ha, hb, hc = a.to_numpy(), b.to_numpy(), c.to_numpy()
        alpha, beta = [(1.0, 1.0), (0.7, 1.3), (-1.0, 0.0)][si % 3]
        if kind >= 2 and si % 2:
            alpha, beta = alpha * (0.7 - 0.9j), beta * (1.3 - 1.1j)
        gemm_dev(opa, opb, alpha, a, b, beta, c)
        torch.cuda.synchronize()
        got = c.to_numpy()
        want = np.array(hc, order="F", copy=True)
        ol.call_f77(ol.oracle_gemm(kind), opa, opb, m, n, k, alpha, np.asfortranarray(ha), ar,
                    np.asfortranarray(hb), br, beta, want, m)
        r = ratio(kind, got, want, alpha, opmat(ha, opa), opmat(hb, opb), beta, hc)
        assert r < RATIO_LIMIT, (kind, ops, m, n, k, r)
        # the same bits again
        c2 = DeviceMatrix(m, n, kind, ld=m + pad).fill_uniform(37 + si)
        gemm_dev(opa, opb, alpha, a, b, beta, c2)
        torch.cuda.synchronize()
        assert np.array_equal(c2.to_numpy(), got)


@pytest.mark.parametrize("kind", range(4))
def test_split_beta_zero_ignores_nan_in_c(kind):
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    m, n, k = 384, 256, 2048
    a = DeviceMatrix(m, k, kind).fill_uniform(1)
    b = DeviceMatrix(k, n, kind).fill_uniform(2)
    c = DeviceMatrix(m, n, kind)
    c.as_strided().fill_(float("nan"))
    gemm_dev("N", "N", 1.0, a, b, 0.0, c)
    torch.cuda.synchronize()
    got = c.to_numpy()
    assert np.isfinite(got).all()
    ha, hb = a.to_numpy(), b.to_numpy()
    want = np.zeros((m, n), dtype=ol.DTYPES[kind], order="F")
    ol.call_f77(ol.oracle_gemm(kind), "N", "N", m, n, k, 1.0, np.asfortranarray(ha), m, np.asfortranarray(hb), k, 0.0,
                want, m)
    assert ratio(kind, got, want, 1.0, ha, hb, 0.0, want) < RATIO_LIMIT


def test_products_on_several_streams_at_once():
    """Each stream's products keep their partial sums in memory of their own: queued side by side
    on four streams, ten products each, every result equals the one computed alone."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    kind, m, n, k = 1, 512, 384, 3072
    a = [DeviceMatrix(m, k, kind).fill_uniform(100 + i) for i in range(4)]
    b = [DeviceMatrix(k, n, kind).fill_uniform(200 + i) for i in range(4)]
    alone = []
    for i in range(4):
        c = DeviceMatrix(m, n, kind).fill_uniform(300 + i)
        gemm_dev("N", "N", 1.0, a[i], b[i], 1.0, c)
        torch.cuda.synchronize()
        alone.append(c.to_numpy())
    streams = [torch.cuda.Stream() for _ in range(4)]
    cs = [[DeviceMatrix(m, n, kind).fill_uniform(300 + i) for _ in range(10)] for i in range(4)]
    torch.cuda.synchronize()
    for rep in range(10):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                gemm_dev("N", "N", 1.0, a[i], b[i], 1.0, cs[i][rep])
    torch.cuda.synchronize()
    for i in range(4):
        for rep in range(10):
            assert np.array_equal(cs[i][rep].to_numpy(), alone[i]), (i, rep)


# (kind, m, n, k): more than two rounds of tiles with a mostly empty last one -- the launcher
# multiplies whole rounds of tiles first and splits the reduction of the remaining tile rows or
# columns.  Too large for the scalar oracle: a sample of rows x columns of C is recomputed on the
# host in double precision from the same operands (same tolerance, the Netlib test ratio).
TAIL_SHAPES = [(1, 2560, 7168, 160), (1, 2688, 7000, 384), (0, 4352, 8192, 256), (3, 1408, 2944, 200),
               (2, 1664, 5120, 192), (1, 7168, 2560, 160)]


@pytest.mark.parametrize("kind,m,n,k", TAIL_SHAPES)
@pytest.mark.parametrize("ops", ["NN", "TN", "NT"])
def test_two_part_launch_sampled(kind, m, n, k, ops):
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    opa, opb = ops
    ar, ac = (m, k) if opa == "N" else (k, m)
    br, bc = (k, n) if opb == "N" else (n, k)
    a = DeviceMatrix(ar, ac, kind).fill_uniform(5)
    b = DeviceMatrix(br, bc, kind).fill_uniform(6)
    c = DeviceMatrix(m, n, kind).fill_uniform(7)
    ha, hb, hc = a.to_numpy(), b.to_numpy(), c.to_numpy()
    alpha, beta = (0.7, 1.3) if kind < 2 else (0.7 - 0.9j, 1.3 - 1.1j)
    gemm_dev(opa, opb, alpha, a, b, beta, c)
    torch.cuda.synchronize()
    got = c.to_numpy()
    rng = np.random.default_rng(m + n + kind)
    # every tile row / column boundary region is hit: the last rows and columns always, the rest at random
    rows = np.unique(np.concatenate([[0, m - 1, m - 200, m - 300], rng.integers(0, m, 96)]))
    cols = np.unique(np.concatenate([[0, n - 1, n - 100, n - 300], rng.integers(0, n, 96)]))
    wide = np.complex128 if kind >= 2 else np.float64
    a_rows = opmat(ha, opa)[rows, :].astype(wide)
    b_cols = opmat(hb, opb)[:, cols].astype(wide)
    c0 = hc[np.ix_(rows, cols)]
    want = alpha * (a_rows @ b_cols) + beta * c0.astype(wide)
    r = ratio(kind, got[np.ix_(rows, cols)], want, alpha, a_rows, b_cols, beta, c0)
    assert r < RATIO_LIMIT, (kind, ops, m, n, k, r)
    # nothing outside the product's own arithmetic: the unsampled part obeys the checksum (C v) = alpha A (B v) + beta C0 v
    v = np.linspace(-1.0, 1.0, n)
    lhs = got.astype(wide) @ v
    rhs = alpha * (opmat(ha, opa).astype(wide) @ (opmat(hb, opb).astype(wide) @ v)) + beta * (hc.astype(wide) @ v)
    scale = np.abs(opmat(ha, opa)).astype(np.float64) @ (np.abs(opmat(hb, opb)).astype(np.float64) @ np.abs(v)) + \
        np.abs(hc).astype(np.float64) @ np.abs(v)
    assert (np.abs(lhs - rhs) <= 64 * ol.EPS[kind] * np.maximum(scale, 1.0) * 4).all()


def test_more_streams_than_buffers_for_the_pieces():
    """Twelve streams multiply at once (the library keeps eight buffers for the pieces of a cut reduction), streams are
    destroyed and created in between: the buffers change hands, first unfenced, then fenced by events, and every
    product stays the bits of the same product run alone.  A stand-alone program (tools/splitk_streams_check.hip, built
    into build/ by __graft_entry__.build()): this process imported PyTorch, whose HIP runtime may lack the stream ids
    the unfenced mode tells streams apart by."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "splitk_streams_check")
    if not os.path.exists(exe):
        pytest.skip("build/splitk_streams_check not built")
    for env_extra in ({}, {"EIGAMD_SPLITK_FENCED": "1"}):
        run = subprocess.run([exe, "12", "6", "1024"], capture_output=True, text=True, timeout=200, env={**os.environ, **env_extra})
        print(run.stdout[-600:])
        assert run.returncode == 0 and "SPLITK STREAMS CHECK OK" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]
