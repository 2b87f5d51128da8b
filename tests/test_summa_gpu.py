"""The multi-GPU GEMM driver with the real HIP local product and the real
block-cyclic device generator, on ONE GPU: 2 and 4 ranks share cuda:0 and exchange
panels over gloo (RCCL refuses several ranks on one device, so the nccl transport
itself is only exercised on a multi-GPU node).  Each rank checks its local part of
C against the product of the globally defined operands."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, kind, n, nb):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(90, exit=True)  # a stuck rank reports where, then exits
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import rng_np
    import eigen_amd  # noqa: F401
    from eigen_amd.summa import DistMatrix, Grid, pgemm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grid = Grid()
        dt = [np.float32, np.float64, np.complex64, np.complex128][kind]
        A = DistMatrix(grid, n, n, nb, nb, kind).fill_uniform(1)
        B = DistMatrix(grid, n, n, nb, nb, kind).fill_uniform(2)
        C = DistMatrix(grid, n, n, nb, nb, kind).fill_uniform(3)
        a, b, c0 = rng_np.matrix(dt, 1, n, n), rng_np.matrix(dt, 2, n, n), rng_np.matrix(dt, 3, n, n)
        # the distributed generator reproduces the single-device matrix
        assert np.array_equal(A.local.to_numpy(), a[np.ix_(A.global_rows(), A.global_cols())])
        for step in range(2):   # two accumulating steps, as bench.py runs them
            pgemm(1.0, A, B, 1.0, C)
        torch.cuda.synchronize()
        want = c0.astype(np.complex128) + 2 * (a.astype(np.complex128) @ b.astype(np.complex128))
        got = C.local.to_numpy()
        ref = want[np.ix_(C.global_rows(), C.global_cols())]
        eps = np.finfo(dt).eps
        gauge = 2 * (np.abs(a) @ np.abs(b)) + np.abs(c0)
        ratio = (np.abs(got - ref) / (eps * gauge[np.ix_(C.global_rows(), C.global_cols())])).max()
        assert ratio < 16.0, (rank, ratio)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_pgemm_real_local_gemm(world):
    mp.spawn(_worker, args=(world, _free_port(), 1, 1024, 128), nprocs=world, join=True)


def _lu_worker(rank, world, port, n):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(90, exit=True)
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import rng_np
    import eigen_amd
    from eigen_amd.device import DeviceMatrix
    from eigen_amd import lu as ealu
    from eigen_amd.plu import ColCyclicMatrix, lu_block_size, pgetrf
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = ColCyclicMatrix(n, lu_block_size(n), eigen_amd.F64).fill_uniform(77)
        full = rng_np.matrix(np.float64, 77, n, n)
        assert np.array_equal(A.local.to_numpy(), full[:, A.global_cols()])
        transp, nswaps, fz = pgetrf(A)
        torch.cuda.synchronize()
        # single-device factorisation of the same matrix: identical pivots, same factors
        S = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(77)
        t1, nb1, fz1 = ealu.getrf_dev(S)
        assert torch.equal(transp.cpu(), t1.cpu()) and nswaps == nb1 and fz == fz1 == -1
        ref = S.to_numpy()[:, A.global_cols()]
        got = A.local.to_numpy()
        assert np.abs(got - ref).max() <= 64 * np.finfo(np.float64).eps * n * np.abs(ref).max(), rank
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_pgetrf_real_device_steps(world):
    """Distributed LU with the library's own panel / swap / solve / GEMM entry points."""
    mp.spawn(_lu_worker, args=(world, _free_port(), 2048), nprocs=world, join=True)
