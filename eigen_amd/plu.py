"""Multi-GPU blocked LU with partial pivoting: the matrix is distributed by column
blocks (1-D block-cyclic, block = the blockSize of the single-device algorithm), one
process per GPU.

It is blocked_lu of the reference (Eigen/src/LU/PartialPivLU.h:429-498) with the loop
body spread over the processes: the owner of column block j factorises the panel
[A11; A21] locally (pivot search needs no communication, every process holds whole
columns), broadcasts the factorised panel and its row transpositions, and every
process applies the transpositions (:479-489), the triangular solve A12 = A11^-1 A12
(:492) and the trailing update A22 -= A21*A12 (:494, the GEMM hot path) to the
column blocks it owns.  Per step there is ONE exchange -- the panel broadcast -- and no
reduction.

torch.distributed is plumbing (broadcast over RCCL/xGMI with backend "nccl"); the
four local operations are the library's device entry points unless a caller injects
others (``ops``), which the CPU tests do to run the schedule under gloo.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .device import DeviceMatrix, _TORCH_OF_KIND
from .summa import device_broadcast, global_indices, numroc


def lu_block_size(size, max_block=256):
    """Block-size rule of blocked_lu (PartialPivLU.h:443-448)."""
    b = (size // 8 // 16) * 16
    return min(max(b, 8), max_block)


class ColCyclicMatrix:
    """n x n matrix; column blocks of width nb dealt cyclically over the ranks of the
    default process group.  `local` holds this rank's columns (n x nloc, ld = n)."""

    def __init__(self, n, nb, kind, device="cuda"):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.n, self.nb, self.kind = n, nb, kind
        self.nloc = numroc(n, nb, self.rank, self.world)
        self.local = DeviceMatrix(n, self.nloc, kind, device=device)

    def global_cols(self):
        return global_indices(self.nloc, self.nb, self.rank, self.world)

    def fill_uniform(self, seed):
        import ctypes
        rc = _lib.lib.eigamd_fill_uniform_cyclic_dev(
            self.kind, ctypes.c_void_p(self.local.ptr), self.n, self.nloc, self.local.ld, seed, max(self.n, 1), 1, 0,
            self.nb, self.world, self.rank, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "eigamd_fill_uniform_cyclic_dev")
        return self

    def set_from_global(self, full):
        part = full[:, self.global_cols()]
        self.local.as_strided().copy_(torch.from_numpy(np.ascontiguousarray(part)))
        return self


class DeviceOps:
    """The four local steps on the library's device entry points.  None of them waits for the
    device: the panel's counters stay in device memory."""

    @staticmethod
    def panel_lu(panel, transp):
        from .lu import getrf_dev_async
        info = torch.empty(3, dtype=torch.int32, device=transp.device)
        getrf_dev_async(panel, transp, info, max_block_size=16)
        # info[2] = 1: a workgroup of the one-launch base case gave up waiting (the device is shared with other
        # work) and the panel is partly garbage; pgetrf carries it along and raises at its single host read.
        return info[0], info[1], info[2]

    @staticmethod
    def laswp(a, transp, k0, k1):
        from .lu import laswp_dev
        laswp_dev(a, transp, k0, k1)

    @staticmethod
    def trsm(l11, a12):
        from .lu import trsm_llnu_dev
        trsm_llnu_dev(l11, a12)

    @staticmethod
    def gemm_minus(a21, a12, a22):
        from .device import gemm_dev
        gemm_dev("N", "N", -1.0, a21, a12, 1.0, a22)


class _PullPanels:
    """Panel transport of pgetrf between processes that own a GPU each: the owner's send slots are opened in every
    process once (inter-process memory handles); when a panel is complete in its slot the owner says so on the
    host-side group, and every other rank copies it out of the owner's HBM on its copy stream (DMA engines, no
    compute unit).  `send(j, ...)` returns a handle whose wait() leaves the panel readable on the current stream."""

    def __init__(self, sbuf, smeta, world):
        from . import transport as tp
        self.tp = tp
        self.group = tp.host_group()
        self.peers = tp.PeerMemory()
        ranks = list(range(world))
        self.wbuf = [tp.Window(t, ranks, self.group, self.peers) for t in sbuf]
        self.wmeta = [tp.Window(t, ranks, self.group, self.peers) for t in smeta]
        self.comm = torch.cuda.Stream()
        self.token = torch.zeros(1, dtype=torch.int32)

    class _Work:
        def __init__(self, fn):
            self.fn = fn

        def wait(self):
            self.fn()
            return True

    def send(self, owner, q, nelem, elem_size, dst, dst_meta, ready, free):
        """Panel in send slot q of `owner` (nelem elements) -> dst / dst_meta on the other ranks.  `ready`: event
        recorded by the owner behind the work that fills the slot; `free`: event behind the last reader of dst on
        this rank (or None) -- the copy waits for that and for nothing else the compute stream has queued, so a
        panel travels beside the bulk update of the step before."""
        me = dist.get_rank()

        def finish():
            if me == owner:
                ready.synchronize()                      # the slot is complete in HBM ...
            dist.broadcast(self.token, src=owner, group=self.group)   # ... and everybody knows
            if me != owner:
                cur = torch.cuda.current_stream()
                if free is not None:
                    self.comm.wait_event(free)
                self.tp.copy_async(dst.data_ptr(), self.wbuf[q].ptr[owner], nelem * elem_size, self.comm)
                self.tp.copy_async(dst_meta.data_ptr(), self.wmeta[q].ptr[owner], dst_meta.numel() * 4, self.comm)
                done = torch.cuda.Event()
                done.record(self.comm)
                cur.wait_event(done)
        return self._Work(finish)

    def close(self):
        self.comm.synchronize()
        dist.barrier(group=self.group)   # nobody frees a slot that a neighbour may still be reading
        self.peers.close()


def pgetrf(A, ops=DeviceOps, broadcast=None, transport=None):
    """In-place P*A = L*U of a ColCyclicMatrix.  Returns (transpositions: int32 tensor of
    n global row indices on A's device, nb_transpositions, first_zero_pivot).

    One panel of look-ahead: as soon as panel j has arrived, the owner of column block j+1
    brings that block up to date, factorises it and starts its broadcast; only then does every
    rank update the rest of its columns with panel j, so panel j+1 is already there (or on its
    way) when that update ends.  An owner factorises into one of two SEND slots of its own
    (panels j, j + 2P, ... of an owner share a slot: by the time it is rewritten every rank has
    factorised a later panel of its own, hence applied -- and before that received -- panel j);
    everybody else receives into one of two receive slots alternating by panel.  The host never
    waits for the device inside the loop with the broadcast transports; the pull transport
    (`transport` = "ipc", see eigen_amd/transport.py) waits once per panel on its owner for the
    panel to be complete.  The zero-pivot and exchange counters are kept in device memory and read
    once at the end."""
    n, nb, P, r = A.n, A.nb, A.world, A.rank
    if nb != lu_block_size(n):
        raise ValueError(f"column block {nb} must equal the LU block size {lu_block_size(n)} for n = {n}")
    dev = A.local.storage.device
    tdt = _TORCH_OF_KIND[A.kind]
    pull = None
    if broadcast is None and P > 1 and dev.type == "cuda":
        from . import transport as tp
        if tp.choose(transport) == "ipc":
            pull = True
    if broadcast is None:
        broadcast = device_broadcast
    transp = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    # factorised panels, contiguous: send slots (this rank's own panels) and receive slots (everybody else's)
    sbuf = [torch.empty(max(n * nb, 1), dtype=tdt, device=dev) for _ in range(2)]
    rbuf = [torch.empty(max(n * nb, 1), dtype=tdt, device=dev) for _ in range(2)] if P > 1 else sbuf
    smeta = [torch.zeros(nb + 3, dtype=torch.int32, device=dev) for _ in range(2)]  # transpositions, first_zero, nswaps, status
    rmeta = [torch.zeros(nb + 3, dtype=torch.int32, device=dev) for _ in range(2)] if P > 1 else smeta
    first_zero = torch.full((), -1, dtype=torch.int32, device=dev)
    nswaps = torch.zeros((), dtype=torch.int32, device=dev)
    status = torch.zeros((), dtype=torch.int32, device=dev)   # non-zero: some panel's factorisation gave up (see DeviceOps.panel_lu)
    nblocks = (n + nb - 1) // nb
    if pull:
        pull = _PullPanels(sbuf, smeta, P)

    def extent(j):
        return j * nb, min(nb, n - j * nb)

    def slots(j):
        """(panel buffer, meta buffer) of panel j on this rank."""
        if r == j % P or P == 1:
            q = (j // P) & 1
            return sbuf[q], smeta[q]
        return rbuf[j & 1], rmeta[j & 1]

    def factor_and_send(j):
        """Panel j: factorised by its owner (its columns are up to date), then on its way to everybody."""
        k, bs = extent(j)
        owner, rows_p = j % P, n - k
        pb, mt = slots(j)
        ready = None
        if r == owner:
            mine = A.local.block(k, (j // P) * nb, rows_p, bs)
            fz, ns, *bad = ops.panel_lu(mine, mt[:bs])          # transpositions relative to row k
            DeviceMatrix(rows_p, bs, A.kind, ld=max(rows_p, 1), storage=pb).as_strided().copy_(mine.as_strided())
            mt[:bs] += k
            mt[nb] = torch.as_tensor(fz, dtype=torch.int32)
            mt[nb + 1] = torch.as_tensor(ns, dtype=torch.int32)
            mt[nb + 2] = torch.as_tensor(bad[0] if bad else 0, dtype=torch.int32)
            if pull:
                ready = torch.cuda.Event()
                ready.record()
        if P == 1:
            return []
        if pull:
            return [pull.send(owner, (j // P) & 1, rows_p * bs, pb.element_size(), pb, mt, ready, freed[j & 1])]
        return [broadcast(pb[:rows_p * bs], owner), broadcast(mt, owner)]

    def apply(j, c0, c1):
        """Transpositions, triangular solve and trailing update of panel j on the local columns [c0, c1)."""
        k, bs = extent(j)
        trows = n - k - bs
        if c1 <= c0:
            return
        cols = A.local.block(0, c0, n, c1 - c0)
        ops.laswp(cols, transp, k, k + bs)
        if trows > 0:
            panel = DeviceMatrix(n - k, bs, A.kind, ld=max(n - k, 1), storage=slots(j)[0])
            a12 = A.local.block(k, c0, bs, c1 - c0)
            ops.trsm(panel.block(0, 0, bs, bs), a12)
            ops.gemm_minus(panel.block(bs, 0, trows, bs), a12, A.local.block(k + bs, c0, trows, c1 - c0))

    freed = [None, None]   # behind the last reader of receive slot 0 / 1
    works = factor_and_send(0) if nblocks else []
    for j in range(nblocks):
        k, bs = extent(j)
        owner = j % P
        for w in works:
            w.wait()
        mt = slots(j)[1]
        fz = mt[nb]
        first_zero = torch.where((first_zero < 0) & (fz >= 0), fz + k, first_zero)
        nswaps = nswaps + mt[nb + 1]
        status = status | mt[nb + 2]
        transp[k:k + bs] = mt[:bs]
        # local columns left of / right of the panel
        owned_upto_j = (j - r) // P + 1 if r <= j else 0          # local blocks with global index <= j
        c_right = min(owned_upto_j * nb, A.nloc)
        c_left_end = (j // P) * nb if r == owner else c_right
        done_upto = c_right                                         # right-hand columns already updated with panel j
        if j + 1 < nblocks:
            if r == (j + 1) % P:                                    # look-ahead: the next panel's columns first
                loc1 = ((j + 1) // P) * nb
                apply(j, loc1, loc1 + extent(j + 1)[1])
                assert loc1 == c_right
                done_upto = loc1 + extent(j + 1)[1]
            works = factor_and_send(j + 1)
        if c_left_end > 0:
            ops.laswp(A.local.block(0, 0, n, c_left_end), transp, k, k + bs)
        apply(j, done_upto, A.nloc)
        if pull:
            freed[j & 1] = torch.cuda.Event()
            freed[j & 1].record()
    fz_host, ns_host, bad_host = (int(v) for v in torch.stack([first_zero, nswaps, status]).cpu())  # the only wait for the device
    if pull:
        pull.close()
    if bad_host:
        raise RuntimeError("pgetrf: a panel's one-launch base case gave up waiting for its workgroups (the device is shared "
                           "with other work); the factors are not valid. Restore the matrix and run again with EIGAMD_LU_FUSED=0")
    return transp[:n], ns_host, fz_host
