"""The multi-GPU GEMM and LU drivers with the real HIP local steps and the real
block-cyclic device generator, on ONE GPU: 2 and 4 ranks share cuda:0 and exchange
panels over gloo (RCCL refuses several ranks on one device, so the nccl transport
itself is only exercised on a multi-GPU node).  Panels travel through a host copy
(host_staged_broadcast); EIGAMD_TEST_DEVICE_BCAST=1 hands the device tensors to the
backend instead, which gloo's device path does not do reliably when the ranks share
a card.  Each rank checks its local part against the globally defined result."""
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


def _bcast():
    from eigen_amd.summa import device_broadcast, host_staged_broadcast
    return device_broadcast if os.environ.get("EIGAMD_TEST_DEVICE_BCAST") else host_staged_broadcast


def _worker(rank, world, port, kind, n, nb, transport=None, backend="gloo"):
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
    torch.cuda.set_stream(torch.cuda.Stream())   # not the legacy default stream
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        grid = Grid()
        if transport == "auto":   # the choice bench.py makes: a probe copy between neighbours, agreed on by all ranks
            from eigen_amd.transport import choose
            transport = choose("auto")
            assert transport in ("ipc", "rccl")
            if rank == 0:
                print(f"transport chosen with {world} rank(s) on backend {backend}: {transport}")
            transport = "rccl" if backend == "nccl" else transport   # (the branch this test is about)
        dt = [np.float32, np.float64, np.complex64, np.complex128][kind]
        A = DistMatrix(grid, n, n, nb, nb, kind).fill_uniform(1)
        B = DistMatrix(grid, n, n, nb, nb, kind).fill_uniform(2)
        C = DistMatrix(grid, n, n, nb, nb, kind).fill_uniform(3)
        a, b, c0 = rng_np.matrix(dt, 1, n, n), rng_np.matrix(dt, 2, n, n), rng_np.matrix(dt, 3, n, n)
        # the distributed generator reproduces the single-device matrix
        assert np.array_equal(A.local.to_numpy(), a[np.ix_(A.global_rows(), A.global_cols())])
        stats = None
        for step in range(2):   # two accumulating steps, as bench.py runs them
            if transport:
                from eigen_amd.transport import PanelStats
                stats = PanelStats()
                pgemm(1.0, A, B, 1.0, C, transport=transport, stats=stats)
            else:
                pgemm(1.0, A, B, 1.0, C, broadcast=_bcast())
        torch.cuda.synchronize()
        if stats is not None:
            t = stats.summary()
            assert t["panels"] == (n + nb - 1) // nb and t["compute_ms"] > 0.0
            if rank == 0:
                print(f"[{transport}, {world} ranks on one GPU] copy stream {t['comm_ms']} ms, products {t['compute_ms']} ms, "
                      f"exposed wait {t['exposed_wait_ms']} ms over {t['panels']} panels")
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


@pytest.mark.parametrize("world,kind,n,nb", [(2, 1, 1024, 128), (4, 1, 1536, 256), (4, 3, 640, 64), (3, 0, 900, 100), (4, 1, 1100, 128)])
def test_pgemm_pull_transport_between_processes(world, kind, n, nb):
    """Every receiver copies the panel piece out of the owner's HBM (inter-process memory handles, a copy stream
    per process): the transport of the one-process-per-GPU driver, here with the processes sharing one GPU."""
    mp.spawn(_worker, args=(world, _free_port(), kind, n, nb, "ipc"), nprocs=world, join=True)


def test_transport_chosen_by_probe_between_two_processes():
    """EIGAMD_SUMMA_TRANSPORT=auto as bench.py uses it: every rank opens its neighbour's probe buffer and copies out of
    it, the ranks agree (all-reduce MIN on the host group), and the product runs over what was chosen."""
    mp.spawn(_worker, args=(2, _free_port(), 1, 1024, 128, "auto"), nprocs=2, join=True)


def test_rccl_branch_with_one_rank():
    """What can run of the RCCL transport on one GPU: backend "nccl" (= RCCL) with a world of one, the collective choice
    of the transport, and the `rccl` branch of the device driver -- the local products on the stream that leaves compute
    units to RCCL's kernels (eigamd_stream_create_cu_mask), ordered behind the copy stream by events.  With one rank
    there is nothing to broadcast; the broadcasts themselves need more than one GPU."""
    mp.spawn(_worker, args=(1, _free_port(), 1, 1536, 256, "auto", "nccl"), nprocs=1, join=True)


def _lu_worker(rank, world, port, n, transport=None, kind=1):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(90, exit=True)
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import rng_np
    import eigen_amd
    from eigen_amd.device import DeviceMatrix
    from eigen_amd.plu import ColCyclicMatrix, DeviceOps, lu_block_size, pgetrf
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())   # not the legacy default stream
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dt = [np.float32, np.float64, np.complex64, np.complex128][kind]
        A = ColCyclicMatrix(n, lu_block_size(n), kind).fill_uniform(77)
        full = rng_np.matrix(dt, 77, n, n)
        assert np.array_equal(A.local.to_numpy(), full[:, A.global_cols()])
        ops = DeviceOps
        if os.environ.get("EIGAMD_TEST_VERBOSE"):
            import time
            t0 = time.time()

            class ops(DeviceOps):   # progress of every local step, to locate a stuck rank
                @staticmethod
                def panel_lu(panel, tr):
                    print(f"[rank {rank}] +{time.time() - t0:.2f}s panel {panel.rows}x{panel.cols}", flush=True)
                    return DeviceOps.panel_lu(panel, tr)

                @staticmethod
                def gemm_minus(a21, a12, a22):
                    torch.cuda.synchronize()
                    print(f"[rank {rank}] +{time.time() - t0:.2f}s update {a22.rows}x{a22.cols}", flush=True)
                    DeviceOps.gemm_minus(a21, a12, a22)
        transp, nswaps, fz = pgetrf(A, ops, transport=transport) if transport else pgetrf(A, ops, _bcast())
        torch.cuda.synchronize()
        # the CPU restatement of blocked_lu on the same matrix: identical pivots, same factors; and the
        # single-device factorisation of the library, run by every rank beside the others on the shared card
        from test_lu_oracle import oracle_lu
        ref_np = np.asfortranarray(full.copy())
        t1, nb1, fz1 = oracle_lu(ref_np)
        assert np.array_equal(transp.cpu().numpy(), t1)
        assert nswaps == nb1 and fz == fz1 == -1
        ref_full = torch.from_numpy(ref_np)
        ref = ref_full.numpy()[:, A.global_cols()]
        got = A.local.to_numpy()
        assert np.abs(got - ref).max() <= 64 * np.finfo(dt).eps * n * np.abs(ref).max(), rank
        from eigen_amd import lu as ealu
        one = np.asfortranarray(full.copy())
        t2, nb2, fz2 = ealu.blocked_lu(one)
        assert np.array_equal(t2, t1) and nb2 == nb1 and fz2 == -1
        assert np.abs(got - one[:, A.global_cols()]).max() <= 64 * np.finfo(dt).eps * n * np.abs(ref).max(), rank
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_pgetrf_real_device_steps(world):
    """Distributed LU with the library's own panel / swap / solve / GEMM entry points."""
    mp.spawn(_lu_worker, args=(world, _free_port(), 1024), nprocs=world, join=True)


@pytest.mark.parametrize("world,n", [(2, 1024), (3, 1100), (4, 1536)])
def test_pgetrf_pull_transport_between_processes(world, n):
    """The same with every panel copied out of its owner's HBM by the receivers (inter-process memory handles)."""
    mp.spawn(_lu_worker, args=(world, _free_port(), n, "ipc"), nprocs=world, join=True)


def test_pgetrf_complex_between_processes():
    """The distributed factorisation for a complex kind (the reference binds its blocked_lu for all four,
    PartialPivLU_LAPACKE.h:74-79): panels pulled between three processes, pivots identical to the oracle's."""
    mp.spawn(_lu_worker, args=(3, _free_port(), 1100, "ipc", 3), nprocs=3, join=True)
