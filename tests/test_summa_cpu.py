"""The multi-GPU GEMM schedule (eigen_amd/summa.py) under gloo on CPU, world sizes 2
and 4: block-cyclic ownership, panel broadcasts, double buffering and the
first-panel beta are exercised with an injected local product (numpy), since the
HIP GEMM needs a GPU.  Sizes are ragged on purpose (partial last blocks, a rank
that owns fewer blocks than its neighbours)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import eigen_amd  # noqa: F401
from eigen_amd import _lib
from eigen_amd.summa import DistMatrix, Grid, grid_shape, numroc, pgemm


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _numpy_local_gemm(alpha, a, b, beta, c):
    av, bv, cv = a.as_strided().numpy(), b.as_strided().numpy(), c.as_strided().numpy()
    cv[...] = alpha * (av @ bv) + beta * cv if a.cols else beta * cv


def _worker(rank, world, port, cases, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grid = Grid()
        dt = [np.float32, np.float64, np.complex64, np.complex128][kind]
        for (M, N, K, nbm, nbn, kb, alpha, beta) in cases:
            rng = np.random.default_rng(1234)  # same global matrices on every rank
            def rnd(r, c):
                x = rng.uniform(-1, 1, (r, c))
                if kind >= 2:
                    x = x + 1j * rng.uniform(-1, 1, (r, c))
                return x.astype(dt)
            a, b, c0 = rnd(M, K), rnd(K, N), rnd(M, N)
            A = DistMatrix(grid, M, K, nbm, kb, kind, device="cpu").set_from_global(a)
            B = DistMatrix(grid, K, N, kb, nbn, kind, device="cpu").set_from_global(b)
            C = DistMatrix(grid, M, N, nbm, nbn, kind, device="cpu").set_from_global(c0)
            pgemm(alpha, A, B, beta, C, local_gemm=_numpy_local_gemm)
            want = alpha * (a.astype(np.complex128) @ b.astype(np.complex128)) + beta * c0
            got = C.local.as_strided().numpy()
            ref = want[np.ix_(C.global_rows(), C.global_cols())]
            tol = (1e-4 if kind in (0, 2) else 1e-12) * max(K, 1)
            assert got.shape == ref.shape
            assert got.size == 0 or np.abs(got - ref).max() <= tol, (rank, M, N, K, np.abs(got - ref).max())
            # every global entry is owned exactly once
            owned = torch.zeros(M * N, dtype=torch.int32)
            idx = (C.global_rows()[:, None] * N + C.global_cols()[None, :]).reshape(-1)
            owned[torch.from_numpy(idx)] = 1
            dist.all_reduce(owned)
            assert bool((owned == 1).all())
    finally:
        dist.destroy_process_group()


CASES = [
    (96, 80, 64, 16, 16, 16, 1.0, 1.0),
    (100, 70, 50, 16, 8, 12, 0.7, 1.3),      # ragged everywhere
    (33, 130, 7, 8, 32, 4, -1.0, 0.0),       # beta == 0, tiny k panels
    (20, 20, 0, 8, 8, 8, 1.0, 1.3),          # k == 0: C = beta*C
    (5, 300, 41, 16, 16, 16, 1.0, 1.0),      # fewer row blocks than grid rows
]


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("kind", [1, 3])
def test_pgemm_gloo(world, kind):
    mp.spawn(_worker, args=(world, _free_port(), CASES, kind), nprocs=world, join=True)


def test_grid_and_numroc():
    assert grid_shape(1) == (1, 1) and grid_shape(2) == (1, 2) and grid_shape(4) == (2, 2) and grid_shape(8) == (2, 4)
    for n, nb, p in [(100, 16, 3), (65536, 2048, 4), (5, 16, 2), (64, 16, 4)]:
        assert sum(numroc(n, nb, i, p) for i in range(p)) == n
