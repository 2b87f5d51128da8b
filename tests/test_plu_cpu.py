"""The multi-GPU LU schedule (eigen_amd/plu.py) under gloo on CPU, world sizes 2 and
3, with the local steps injected (panel factorisation = the LU oracle with
maxBlockSize 16, swaps / solve / update in numpy): the distributed result must have
exactly the pivots of the single-process algorithm and the same factors to rounding."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
import eigen_amd  # noqa: F401
from eigen_amd.plu import ColCyclicMatrix, lu_block_size, pgetrf
from test_lu_oracle import oracle_lu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class NumpyOps:
    @staticmethod
    def panel_lu(panel, transp):
        a = panel.as_strided().numpy()            # column-major view, strides (8, 8*ld)
        t = np.zeros(max(min(a.shape), 1), dtype=np.int32)
        nb = ctypes.c_int()
        r = ol.oracle.oracle_blocked_lu_d(a.shape[0], a.shape[1], a.ctypes.data, panel.ld, t.ctypes.data,
                                          ctypes.byref(nb), 16)
        transp[:min(a.shape)] = torch.from_numpy(t[:min(a.shape)])
        return r, nb.value

    @staticmethod
    def laswp(a, transp, k0, k1):
        v = a.as_strided().numpy()
        for i in range(k0, k1):
            p = int(transp[i])
            if p != i:
                v[[i, p], :] = v[[p, i], :]

    @staticmethod
    def trsm(l11, a12):
        l = np.tril(l11.as_strided().numpy(), -1) + np.eye(l11.rows)
        b = a12.as_strided().numpy()
        for i in range(l11.rows):                 # forward substitution, column (saxpy) order
            b[i + 1:, :] -= np.outer(l[i + 1:, i], b[i, :])

    @staticmethod
    def gemm_minus(a21, a12, a22):
        c = a22.as_strided().numpy()
        c -= a21.as_strided().numpy() @ a12.as_strided().numpy()


def _worker(rank, world, port, sizes):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for n in sizes:
            rng = np.random.default_rng(n)
            full = np.asfortranarray(rng.uniform(-1, 1, (n, n)))
            A = ColCyclicMatrix(n, lu_block_size(n), 1, device="cpu").set_from_global(full)
            transp, nswaps, fz = pgetrf(A, ops=NumpyOps)
            want = full.copy(order="F")
            t_ref, nb_ref, r_ref = oracle_lu(want)
            assert np.array_equal(transp.numpy(), t_ref), (rank, n)
            assert nswaps == nb_ref and fz == r_ref == -1
            got = A.local.as_strided().numpy()
            ref = want[:, A.global_cols()]
            assert np.abs(got - ref).max() <= 64 * np.finfo(np.float64).eps * n * np.abs(want).max(), (rank, n)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pgetrf_gloo(world):
    mp.spawn(_worker, args=(world, _free_port(), [40, 200, 300, 520]), nprocs=world, join=True)


def test_block_size_rule():
    assert lu_block_size(32768) == 256 and lu_block_size(1024) == 128 and lu_block_size(200) == 16 and lu_block_size(40) == 8
