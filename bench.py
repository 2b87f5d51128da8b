#!/usr/bin/env python3
"""Headline benchmark: dense GEMM throughput of the HIP path on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
      one step = one C += A*B, double precision, 16384^3, column-major, operands
      already in HBM (mirrors bench/bench_gemm.cpp:161-165,243-246 of the reference:
      A, B random in [-1,1), C accumulates).
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
      one step = the same product at 65536^3, A/B/C distributed 2-D block-cyclically
      over the N GPUs (eigen_amd/summa.py: panel pieces pulled out of their owners' HBM
      by the DMA engines, or RCCL broadcasts; one local GEMM per k-panel).  Fixed total
      work => "scaling": "strong".  The line then carries "multi_gpu": per rank the time
      on the copy stream, in the local products, and waiting for a panel.

Prints ONE JSON line on rank 0.  `value` is whole-job TFLOP/s (2*n^3*steps / wall
time, max over ranks, barrier + device synchronize on both sides).  `roofline` is
the dominant kernel (the fp64 MFMA GEMM) priced against the fp64 matrix peak, from
HIP events recorded around every launch on the launch stream.  `cpu_baseline` is
Eigen's own multithreaded operator* product (oracle/_ref, built from the
reference) timed on this host on a bounded sample -- context, not a target.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Dense MFMA peaks (TFLOP/s).  f32: 64 FLOP/clk/SIMD * 4 SIMD * 256 CU * 2.4 GHz.
# f64: v_mfma_f64_16x16x4_f64 issues at half that rate (32 FLOP/clk/SIMD).
PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3, "c64": 78.6, "c32": 157.3}
KIND = {"f32": 0, "f64": 1, "c32": 2, "c64": 3}
BLAS_LETTER = {"f32": "s", "f64": "d", "c32": "c", "c64": "z"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--size", type=int, default=None, help="n of the n^3 product (default 16384 on 1 GPU, 65536 on >1)")
    ap.add_argument("--dtype", default="f64", choices=list(KIND))
    ap.add_argument("--kb", type=int, default=2048, help="block size of the block-cyclic layout (= k-panel width)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the sgemm / complex side measurements")
    ap.add_argument("--cpu-size", type=int, default=8192)
    ap.add_argument("--lu", type=int, default=0, metavar="N",
                    help="with >1 GPU: also time the distributed LU of an N x N double matrix (extra key)")
    return ap.parse_args()


class TimedGemm:
    """Wraps gemm_dev: records a HIP event pair around every launch on the launch stream."""

    def __init__(self):
        self.events = []
        self.enabled = False

    def __call__(self, alpha, a, b, beta, c):
        import torch
        from eigen_amd.device import gemm_dev
        if self.enabled:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gemm_dev("N", "N", alpha, a, b, beta, c)
            e1.record()
            self.events.append((e0, e1, 2.0 * c.rows * c.cols * a.cols * (4 if c.kind >= 2 else 1)))
        else:
            gemm_dev("N", "N", alpha, a, b, beta, c)

    def summary(self):
        ms = [e0.elapsed_time(e1) for e0, e1, _ in self.events]
        flops = [f for _, _, f in self.events]
        return ms, flops


def sampled_ratio(kind, n, c_rows_cols_getter, steps_done, nsamp=24):
    """Test ratio of sampled C entries after `steps_done` accumulations of A*B onto C0
    (operands regenerated on the host from the counter-based generator)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rng_np
    dt = [np.float32, np.float64, np.complex64, np.complex128][kind]
    rng = np.random.default_rng(99)
    rows = np.unique(np.concatenate([[0, n - 1], rng.integers(0, n, nsamp)]))
    cols = np.unique(np.concatenate([[0, n - 1], rng.integers(0, n, nsamp)]))
    kk = np.arange(n)
    a_rows = rng_np.entries(dt, 1, rows[:, None], kk[None, :])
    b_cols = rng_np.entries(dt, 2, kk[:, None], cols[None, :])
    c0 = rng_np.entries(dt, 3, rows[:, None], cols[None, :])
    wide = np.complex128 if kind >= 2 else np.float64
    prod = a_rows.astype(wide) @ b_cols.astype(wide)
    want = c0.astype(wide) + steps_done * prod
    got, mask = c_rows_cols_getter(rows, cols)  # got: values, mask: which (row, col) are available
    abs1 = (lambda z: np.abs(z.real) + np.abs(z.imag))
    gauge = steps_done * (abs1(a_rows).astype(np.float64) @ abs1(b_cols).astype(np.float64)) + abs1(c0)
    eps = 2.0 ** -23 if kind in (0, 2) else 2.0 ** -52
    r = abs1(got.astype(np.complex128) - want) / (eps * gauge)
    r = r[mask]
    return float(r.max()) if r.size else 0.0


def cpu_baseline(args):
    """Eigen's own CPU product on this host: c.noalias() += a*b, best of 2 after a
    warm-up, through the reference build under oracle/_ref (multithreaded, OpenMP)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import oracle_lib as ol
    except Exception as e:  # oracle not built
        return {"value": None, "unit": "TFLOP/s", "cores": 0, "kind": "reference", "sample": f"unavailable: {e}"}
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    ref = ol.load_ref()
    n = args.cpu_size
    if ref is not None:
        ref.ref_set_threads(threads)
        secs = ref.ref_bench_gemm(1, n, n, n, 2)
        # the reference's own default case (bench/bench_gemm.cpp:179: 2048^3, double) beside it
        secs_2048 = ref.ref_bench_gemm(1, 2048, 2048, 2048, 3)
        return {"eigen_cpu_2048_tflops": round(2.0 * 2048 ** 3 / secs_2048 / 1e12, 4), "value": round(2.0 * n ** 3 / secs / 1e12, 4), "unit": "TFLOP/s", "cores": int(ref.ref_num_threads()),
                "kind": "reference",
                "sample": f"Eigen operator* (c.noalias()+=a*b) double {n}^3, best of 2, {ref.variant} "
                          f"[{ref.ref_simd().decode()}], {2.0 * n ** 3 / (2.0 * 16384 ** 3):.4f} of one step's flops",
                "seconds": round(secs, 4)}
    # fall back to the C restatement (single thread) on a smaller sample
    import numpy as np
    import eigen_amd
    n = 1024
    a = eigen_amd.fill_uniform_host(np.float64, n, n, 1)
    b = eigen_amd.fill_uniform_host(np.float64, n, n, 2)
    c = np.ones((n, n), order="F")
    t0 = time.perf_counter()
    ol.call_f77(ol.oracle_gemm(1), "N", "N", n, n, n, 1.0, a, n, b, n, 1.0, c, n)
    secs = time.perf_counter() - t0
    return {"value": round(2.0 * n ** 3 / secs / 1e12, 5), "unit": "TFLOP/s", "cores": 1, "kind": "port",
            "sample": f"C oracle dgemm {n}^3, one run", "seconds": round(secs, 4)}


def side_measurement(dtype, n, steps):
    """Device-resident C += A*B for another scalar kind; returns TFLOP/s (best step)."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    kind = KIND[dtype]
    a = DeviceMatrix(n, n, kind).fill_uniform(1)
    b = DeviceMatrix(n, n, kind).fill_uniform(2)
    c = DeviceMatrix(n, n, kind).fill_uniform(3)
    gemm_dev("N", "N", 1.0, a, b, 1.0, c)
    best = 1e30
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gemm_dev("N", "N", 1.0, a, b, 1.0, c)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    flops = (8.0 if kind >= 2 else 2.0) * n ** 3
    return round(flops / best / 1e9, 2)


def trsm_measurement(n, letter="d", form="LLNN"):
    """?trsm (side, uplo, trans, diag = `form`) with an n x n triangle and n right-hand sides resident in HBM
    (eigamd_trsm_dev); n^3 flop (4 n^3 for the complex kinds).  The block updates run on the GEMM kernel."""
    import torch
    import eigen_amd
    from eigen_amd.device import DeviceMatrix, trsm_dev
    kind = {"s": eigen_amd.F32, "d": eigen_amd.F64, "c": eigen_amd.C32, "z": eigen_amd.C64}[letter]
    a = DeviceMatrix(n, n, kind).fill_uniform(5)
    a.as_strided().mul_(2.0 / n)
    a.as_strided().diagonal().fill_(1.5)
    b = DeviceMatrix(n, n, kind)
    best = 1e30
    for _ in range(3):
        b.fill_uniform(6)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        trsm_dev(form[0], form[1], form[2], form[3], 1.0, a, b)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    flops = (4.0 if letter in "cz" else 1.0) * float(n) ** 3
    return {f"{letter}trsm_{form.lower()}_{n}_tflops": round(flops / best / 1e9, 2)}


def syrk_measurement(n):
    """dsyrk('L','N') with an n x n triangle and k = n resident in HBM (eigamd_syrk_dev);
    n^2 k flop.  One GEMM launch: tiles outside the triangle leave at once, diagonal tiles write
    through a row/column predicate in the epilogue."""
    import torch
    import eigen_amd
    from eigen_amd.device import DeviceMatrix, syrk_dev
    a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(7)
    c = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(8)
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        syrk_dev("L", "N", 1.0, a, 1.0, c)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return {f"dsyrk_ln_{n}_tflops": round(float(n) ** 3 / best / 1e9, 2)}


def trmm_measurement(n):
    """dtrmm('L','L','N','N') with an n x n triangle and n columns resident in HBM
    (eigamd_trmm_dev); n^3 flop (the useful half of the dense product)."""
    import torch
    import eigen_amd
    from eigen_amd.device import DeviceMatrix, trmm_dev
    a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(9)
    b = DeviceMatrix(n, n, eigen_amd.F64)
    best = 1e30
    for _ in range(3):
        b.fill_uniform(10)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        trmm_dev("L", "L", "N", "N", 1.0, a, b)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return {f"dtrmm_llnn_{n}_tflops": round(float(n) ** 3 / best / 1e9, 2)}


def mixed_measurement(n):
    """Complex-double A times real-double B (eigamd_gemm_mixed_dev), C += A*B resident in HBM:
    4 n^3 flop (a complex x real multiply-add is 2 multiplications and 2 additions).  With a
    plain A and real alpha it is one launch of the real kernel on the (re, im)-interleaved rows."""
    import torch
    import eigen_amd
    from eigen_amd.device import DeviceMatrix, gemm_mixed_dev
    a = DeviceMatrix(n, n, eigen_amd.C64).fill_uniform(11)
    b = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(12)
    c = DeviceMatrix(n, n, eigen_amd.C64).fill_uniform(13)
    gemm_mixed_dev("N", "N", 1.0, a, b, 1.0, c)
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gemm_mixed_dev("N", "N", 1.0, a, b, 1.0, c)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return {f"zgemm_complex_x_real_{n}_tflops": round(4.0 * float(n) ** 3 / best / 1e9, 2)}


def lu_measurement(n):
    """PartialPivLU-equivalent factorisation (eigamd_getrf_dev) of an n x n double matrix
    resident in HBM; 2/3 n^3 flop.  The trailing updates run on the GEMM kernel.  What was timed
    is checked: transpositions in range, |l_ij| <= 1, and the reference's own acceptance
    criterion (test/lu.cpp:168, P*A = L*U to 8 * eps * n) on sampled block rows."""
    import numpy as np
    import torch
    import eigen_amd
    from eigen_amd.device import DeviceMatrix
    from eigen_amd import lu as ealu
    a = DeviceMatrix(n, n, eigen_amd.F64)
    best = 1e30
    for _ in range(2):
        a.fill_uniform(77)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        transp, _, first_zero = ealu.getrf_dev(a)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    t = transp.cpu().numpy()
    ok = first_zero == -1 and bool((t >= np.arange(n)).all() and (t < n).all())
    perm = np.arange(n)
    for i, p in enumerate(t):
        if p != i:
            perm[i], perm[p] = perm[p], perm[i]
    lu = a.as_strided()
    a0 = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(77).as_strided()
    u = torch.triu(lu)
    cols = torch.arange(n, device="cuda")
    worst, lmax, block = 0.0, 0.0, 128
    for r0 in sorted({0, n // 3, n // 2 + 64, n - block}):
        rows = torch.arange(r0, r0 + block, device="cuda")
        lr = torch.where(cols[None, :] < rows[:, None], lu[rows], torch.zeros((), dtype=lu.dtype, device="cuda"))
        lmax = max(lmax, float(lr.abs().max()))
        lr[torch.arange(block, device="cuda"), rows] = 1.0
        pa = a0[torch.as_tensor(perm[r0:r0 + block], device="cuda")]
        worst = max(worst, float(torch.linalg.norm(pa - lr @ u) / torch.linalg.norm(pa)))
    eps = 2.0 ** -52
    out = {f"lu_{n}_f64_seconds": round(best, 4), f"lu_{n}_f64_tflops": round(2.0 / 3.0 * n ** 3 / best / 1e12, 2),
           f"lu_{n}_residual_over_eps_n": round(worst / (eps * n), 4), f"lu_{n}_residual_limit": 8.0,
           f"lu_{n}_max_abs_l": round(lmax, 6), f"lu_{n}_checks_ok": bool(ok and worst <= 8 * eps * n and lmax <= 1.0)}
    # the solve that consumes the factors (eigamd_getrs_dev: row exchanges + two triangular solves, 2 n^2 nrhs flop),
    # checked on sampled rows of A: ||A(rows, :) X - B(rows, :)|| <= 8 eps n ||A(rows, :)|| ||X||
    del u
    nrhs = 2048
    rows = torch.as_tensor(np.random.default_rng(5).choice(n, 64, replace=False), device="cuda")
    a_rows = a0[rows].clone()
    del a0
    b = DeviceMatrix(n, nrhs, eigen_amd.F64)
    sbest = 1e30
    for _ in range(2):
        b.fill_uniform(22)
        b_rows = b.as_strided()[rows].clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ealu.getrs_dev(eigen_amd.OP_N, a, transp, b)
        e1.record()
        torch.cuda.synchronize()
        sbest = min(sbest, e0.elapsed_time(e1) * 1e-3)
    x = b.as_strided()
    res = float(torch.linalg.norm(a_rows @ x - b_rows) / (torch.linalg.norm(a_rows) * torch.linalg.norm(x)))
    out.update({f"lu_solve_{n}x{nrhs}_f64_seconds": round(sbest, 4),
                f"lu_solve_{n}x{nrhs}_f64_tflops": round(2.0 * n * n * nrhs / sbest / 1e12, 2),
                f"lu_solve_{n}_backward_error_over_eps_n": round(res / (eps * n), 4),
                f"lu_solve_{n}_checks_ok": bool(res <= 8 * eps * n)})
    return out


def lu_sweep_measurements():
    """eigamd_getrf_dev below the configured size, where the factorisation is the panel chain alone, and for the other
    kinds (best of 2; 2/3 n^3 flop, 8/3 n^3 for the complex kinds), each with the residual check of test/lu.cpp:168 on
    sampled block rows done in complex128 / float64 on the device."""
    import numpy as np
    import torch
    import eigen_amd
    from eigen_amd.device import DeviceMatrix
    from eigen_amd import lu as ealu
    out = {}
    for dtype, n in (("f64", 8192), ("f64", 16384), ("f32", 8192), ("c64", 8192), ("c32", 8192)):
        kind = KIND[dtype]
        a = DeviceMatrix(n, n, kind)
        best = 1e30
        for _ in range(2):
            a.fill_uniform(77)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            transp, _, first_zero = ealu.getrf_dev(a)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        t = transp.cpu().numpy()
        perm = np.arange(n)
        for i, p in enumerate(t):
            if p != i:
                perm[i], perm[p] = perm[p], perm[i]
        wide = torch.complex128 if kind >= 2 else torch.float64
        lu = a.as_strided()
        a0 = DeviceMatrix(n, n, kind).fill_uniform(77).as_strided()
        u = torch.triu(lu).to(wide)
        cols = torch.arange(n, device="cuda")
        worst, block = 0.0, 128
        for r0 in (0, n // 2 + 64, n - block):
            rows = torch.arange(r0, r0 + block, device="cuda")
            lr = torch.where(cols[None, :] < rows[:, None], lu[rows], torch.zeros((), dtype=lu.dtype, device="cuda")).to(wide)
            lr[torch.arange(block, device="cuda"), rows] = 1.0
            pa = a0[torch.as_tensor(perm[r0:r0 + block], device="cuda")].to(wide)
            worst = max(worst, float(torch.linalg.norm(pa - lr @ u) / torch.linalg.norm(pa)))
        eps = 2.0 ** (-52 if dtype in ("f64", "c64") else -23)
        name = f"{BLAS_LETTER[dtype]}getrf_{n}"
        out[f"{name}_ms"] = round(best * 1e3, 2)
        out[f"{name}_tflops"] = round((8.0 if kind >= 2 else 2.0) / 3.0 * n ** 3 / best / 1e12, 2)
        out[f"{name}_checks_ok"] = bool(first_zero == -1 and (t >= np.arange(n)).all() and (t < n).all() and worst <= 8 * eps * n)
        del a, a0, u, lu
    return out


def variant_measurements():
    """Device-resident C += op(A)*op(B) at the configured sizes for the transpose / conjugate forms
    (TFLOP/s, best of 3): dgemm / sgemm 16384^3 NN TN NT TT, zgemm / cgemm 8192^3 NN TN NC CC."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    out = {}
    for dtype, n, forms in (("f64", 16384, ("TN", "NT", "TT")), ("f32", 16384, ("TN", "NT", "TT")),
                            ("c64", 8192, ("TN", "NC", "CC")), ("c32", 8192, ("TN", "NC", "CC"))):
        kind = KIND[dtype]
        a = DeviceMatrix(n, n, kind).fill_uniform(1)
        b = DeviceMatrix(n, n, kind).fill_uniform(2)
        c = DeviceMatrix(n, n, kind).fill_uniform(3)
        flops = (8.0 if kind >= 2 else 2.0) * n ** 3
        for f in forms:
            gemm_dev(f[0], f[1], 1.0, a, b, 1.0, c)
            best = 1e30
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gemm_dev(f[0], f[1], 1.0, a, b, 1.0, c)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            out[f"{BLAS_LETTER[dtype]}gemm_{n}_{f}_tflops"] = round(flops / best / 1e9, 2)
        del a, b, c
    return out


def mid_size_measurements():
    """Device-resident square products of a few rounds of tiles, 20 launches back to back after 20 untimed ones
    (a lone sub-millisecond launch measures the clocks coming up, not the kernel): TFLOP/s."""
    import torch
    from eigen_amd.device import DeviceMatrix, gemm_dev
    out = {}
    for dtype in ("f64", "f32"):
        kind = KIND[dtype]
        for n in (1024, 2048, 2560, 3072, 4096):
            a = DeviceMatrix(n, n, kind).fill_uniform(1)
            b = DeviceMatrix(n, n, kind).fill_uniform(2)
            c = DeviceMatrix(n, n, kind).fill_uniform(3)
            for _ in range(20):
                gemm_dev("N", "N", 1.0, a, b, 0.5, c)
            best = 1e30
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    gemm_dev("N", "N", 1.0, a, b, 0.5, c)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            out[f"{BLAS_LETTER[dtype]}gemm_{n}_batch20_tflops"] = round(2.0 * n ** 3 / best / 1e9, 2)
            del a, b, c
    return out


def measured_traffic(dtype, n):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    passes (FETCH_SIZE doubled as gfx950 requires, plus WRITE_SIZE); None when no
    measurement for this workload is on file."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
        entry = table.get(f"{dtype}_{n}")
        return (entry["bytes_per_launch"], entry["source"]) if entry else (None, None)
    except (OSError, ValueError, KeyError):
        return (None, None)


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import eigen_amd  # noqa: F401  (raises if the HIP library is missing)
    from eigen_amd import _lib
    from eigen_amd.device import DeviceMatrix
    from eigen_amd.summa import DistMatrix, Grid, pgemm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world
    # EIGAMD_BENCH_REHEARSAL=1: run the multi-rank code path on fewer GPUs than ranks (ranks
    # share devices, gloo instead of RCCL for the host-side collectives, panels by the pull
    # transport).  For checking the driver logic on a one-GPU box only; its numbers mean nothing.
    rehearsal = os.environ.get("EIGAMD_BENCH_REHEARSAL") == "1"
    torch.cuda.set_device(local_rank % torch.cuda.device_count() if rehearsal else local_rank)
    if _lib.lib.eigamd_device_ok() != 1:
        raise SystemExit("bench.py needs an MI355X: " + _lib.lib.eigamd_last_error().decode())
    transport = None
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # How panel pieces travel between the ranks (eigen_amd/transport.py): EIGAMD_SUMMA_TRANSPORT = ipc | rccl | auto.
        # auto takes the pull transport (receivers copy out of the owner's HBM with the DMA engines) when every rank can
        # open its neighbours' memory, else RCCL broadcasts with the local products on a stream that leaves compute units
        # free for them.  A rehearsal has no RCCL (its ranks share devices).
        from eigen_amd import transport as tp
        transport = "ipc" if rehearsal else tp.choose(None)

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    n = args.size or (16384 if world == 1 else 65536)
    steps = args.steps if args.steps is not None else (10 if world == 1 else 3)
    warmup = args.warmup if args.warmup is not None else (2 if world == 1 else 1)
    kind = KIND[args.dtype]
    cplx = kind >= 2
    flops_per_step = (8.0 if cplx else 2.0) * float(n) ** 3
    timed = TimedGemm()

    if world == 1:
        A = DeviceMatrix(n, n, kind).fill_uniform(1)
        B = DeviceMatrix(n, n, kind).fill_uniform(2)
        C = DeviceMatrix(n, n, kind).fill_uniform(3)
        step = lambda: timed(1.0, A, B, 1.0, C)
        layout = "single device, column-major, ld = n"
    else:
        grid = Grid()
        nb = args.kb
        A = DistMatrix(grid, n, n, nb, nb, kind).fill_uniform(1)
        B = DistMatrix(grid, n, n, nb, nb, kind).fill_uniform(2)
        C = DistMatrix(grid, n, n, nb, nb, kind).fill_uniform(3)
        panel_stats = []

        def step():
            st = tp.PanelStats() if timed.enabled else None
            pgemm(1.0, A, B, 1.0, C, local_gemm=timed, transport=transport, stats=st)
            if st is not None:
                panel_stats.append(st)
        how = ("every receiver copies the piece out of its owner's HBM (inter-process memory handles, DMA engines)"
               if transport == "ipc" else
               f"RCCL broadcasts on row / column communicators, local products on a stream that leaves "
               f"{os.environ.get('EIGAMD_SUMMA_RCCL_CUS', '32')} compute units free")
        layout = f"2-D block-cyclic {grid.nprow}x{grid.npcol}, block {nb}, SUMMA one panel ahead; {how}"

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timed.enabled = True
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed.enabled = False
    if world > 1:
        elapsed = max_over_ranks(elapsed)

    # parity of what was just timed: sampled entries of C against the host
    done = warmup + steps
    if world == 1:
        def getter(rows, cols):
            sub = C.as_strided()[torch.as_tensor(rows, device="cuda")][:, torch.as_tensor(cols, device="cuda")]
            return sub.cpu().numpy(), np.ones((len(rows), len(cols)), dtype=bool)
    else:
        def getter(rows, cols):
            gr, gc = C.global_rows(), C.global_cols()
            rpos = {int(g): i for i, g in enumerate(gr)}
            cpos = {int(g): i for i, g in enumerate(gc)}
            got = np.zeros((len(rows), len(cols)), dtype=[np.float32, np.float64, np.complex64, np.complex128][kind])
            mask = np.zeros((len(rows), len(cols)), dtype=bool)
            loc = C.local.as_strided()
            for i, r in enumerate(rows):
                for j, c in enumerate(cols):
                    if int(r) in rpos and int(c) in cpos:
                        got[i, j] = loc[rpos[int(r)], cpos[int(c)]].item()
                        mask[i, j] = True
            return got, mask
    ratio = sampled_ratio(kind, n, getter, done)
    if world > 1:
        ratio = max_over_ranks(ratio)

    lu_extra = None
    if world > 1 and args.lu:
        from eigen_amd.plu import ColCyclicMatrix, lu_block_size, pgetrf
        warm = ColCyclicMatrix(2048, lu_block_size(2048), KIND["f64"]).fill_uniform(78)   # (one-time set-up -- windows, streams, scratch -- is not the factorisation)
        pgetrf(warm, transport=transport)
        L = ColCyclicMatrix(args.lu, lu_block_size(args.lu), KIND["f64"]).fill_uniform(77)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        _, _, fz = pgetrf(L, transport=transport)
        torch.cuda.synchronize()
        dist.barrier()
        lu_secs = max_over_ranks(time.perf_counter() - t0)
        lu_extra = {f"lu_{args.lu}_f64_seconds": round(lu_secs, 4),
                    f"lu_{args.lu}_f64_tflops": round(2.0 / 3.0 * args.lu ** 3 / lu_secs / 1e12, 2),
                    "lu_layout": f"1-D column block-cyclic over {world} GPUs, block {lu_block_size(args.lu)}"}

    # Where every rank's time went, so that a scaling run explains its own efficiency: time on the copy stream,
    # time in the local products, and how long the compute stream stood waiting for a panel (HIP events per panel).
    multi = None
    if world > 1:
        mine = {"rank": rank, "device": torch.cuda.current_device(), "comm_ms": 0.0, "compute_ms": 0.0, "exposed_wait_ms": 0.0,
                "exposed_wait_ms_per_panel_max": []}
        for st in panel_stats:
            t = st.summary()
            for key in ("comm_ms", "compute_ms", "exposed_wait_ms"):
                mine[key] = round(mine[key] + t[key], 3)
            w = t["exposed_wait_ms_per_panel"]
            cur = mine["exposed_wait_ms_per_panel_max"]
            mine["exposed_wait_ms_per_panel_max"] = [max(a, b) for a, b in zip(w, cur)] if cur else w
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=tp.host_group())
        multi = {"transport": transport, "ranks": world, "rccl_ranks": world if dist.get_backend() == "nccl" else 0,
                 "grid": [grid.nprow, grid.npcol], "panels_per_step": (n + nb - 1) // nb, "steps_timed": steps,
                 "wall_ms": round(elapsed * 1e3, 3), "per_rank": everyone}

    ms, fl = timed.summary()
    kernel_tflops = sum(fl) / (sum(ms) * 1e-3) / 1e12 if ms else 0.0
    peak = PEAK_TFLOPS[args.dtype]
    value = flops_per_step * steps / elapsed / 1e12

    if rank == 0:
        out = {
            "metric": "GEMM TFLOP/s",
            "value": round(value, 3),
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{BLAS_LETTER[args.dtype]}gemm {n}^3 C += A*B (alpha=1, beta=1, NN, column-major)",
                       "layout": layout, "n": n, "global_flops_per_step": flops_per_step},
            "pct_of_peak": round(100.0 * value / (peak * world), 2),
            "parity_max_test_ratio": round(ratio, 3),
            "parity_limit": 16.0,
            "roofline": {"bound": "mfma", "achieved": round(kernel_tflops, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(kernel_tflops / peak, 4),
                         # (the counters were collected for the single-device launch; a rank's local products are other launches)
                         "traffic": measured_traffic(args.dtype, n)[0] if world == 1 else None,
                         "traffic_source": measured_traffic(args.dtype, n)[1] if world == 1 else None,
                         "kernel": f"eigamd::gemm_kernel<{args.dtype}, A m-contiguous, B k-contiguous>",
                         "algorithmic_bytes_per_launch": 4.0 * n * n * {"f32": 4, "f64": 8, "c32": 8, "c64": 16}[args.dtype]
                         if world == 1 else None,
                         "launches": len(ms), "avg_launch_ms": round(sum(ms) / max(len(ms), 1), 4),
                         "flops_per_launch": fl[0] if fl else 0},
        }
        if multi:
            out["multi_gpu"] = multi
        if lu_extra:
            out["extra"] = lu_extra
        if world == 1:
            if not args.no_extra:
                out["extra"] = {
                    "sgemm_16384_tflops": side_measurement("f32", 16384, 3),
                    "zgemm_8192_tflops": side_measurement("c64", 8192, 3),
                    "cgemm_8192_tflops": side_measurement("c32", 8192, 3),
                }
                out["extra"].update(lu_measurement(32768))
                out["extra"].update(lu_sweep_measurements())
                out["extra"].update(trsm_measurement(16384))
                out["extra"].update(trsm_measurement(8192))
                out["extra"].update(trsm_measurement(4096))
                out["extra"].update(trsm_measurement(4096, "d", "RLNN"))
                out["extra"].update(trsm_measurement(4096, "s"))
                out["extra"].update(trsm_measurement(2048, "z"))
                out["extra"].update(syrk_measurement(16384))
                out["extra"].update(trmm_measurement(16384))
                out["extra"].update(mixed_measurement(8192))
                out["extra"].update(variant_measurements())
                out["extra"].update(mid_size_measurements())
                out["extra"]["sgemm_pct_of_peak"] = round(100 * out["extra"]["sgemm_16384_tflops"] / PEAK_TFLOPS["f32"], 2)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
