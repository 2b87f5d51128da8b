"""Multi-GPU GEMM: C = alpha*A*B + beta*C with A, B, C distributed 2-D
block-cyclically over a process grid, one process per GPU.

The reference has no distributed layer; its only parallel scheme is the OpenMP
split of general_matrix_matrix_product (Eigen/src/Core/products/
GeneralMatrixMatrix.h:86-158, Parallelizer.h:101-176): every thread owns a column
slab of C and all share one packed A panel per k-block.  The multi-GPU analogue
is SUMMA: per k-panel, the grid column that owns A(:, panel) broadcasts it along
each grid row, the grid row that owns B(panel, :) broadcasts it along each grid
column, and every process runs ONE local GEMM on its whole local C.  K is never
split, so there is no reduction -- only the two panel broadcasts, which run on
the communication backend (RCCL over xGMI with backend "nccl") one panel ahead of
the local GEMM.

torch.distributed is plumbing here (process groups, broadcast); the local
product is the library's device GEMM unless a caller injects another
(``local_gemm``), which the CPU tests do to run the schedule under gloo.
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .device import DeviceMatrix, _TORCH_OF_KIND


def grid_shape(world):
    """nprow x npcol with nprow <= npcol, as square as the rank count allows
    (2 -> 1x2, 4 -> 2x2, 8 -> 2x4)."""
    r = int(math.isqrt(world))
    while world % r:
        r -= 1
    return r, world // r


def numroc(n, nb, p, nprocs):
    """Rows (or columns) of an n-long dimension, blocked by nb and dealt cyclically
    over nprocs processes, that land on process p."""
    nblocks = n // nb
    base = (nblocks // nprocs) * nb
    extra = nblocks % nprocs
    if p < extra:
        return base + nb
    if p == extra:
        return base + n % nb
    return base


def global_indices(nloc, nb, p, nprocs):
    l = np.arange(nloc, dtype=np.int64)
    return (l // nb) * nb * nprocs + p * nb + l % nb


class Grid:
    """Process grid over the default process group; ranks are laid out row-major."""

    def __init__(self, nprow=None, npcol=None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        if nprow is None:
            nprow, npcol = grid_shape(self.world)
        if nprow * npcol != self.world:
            raise ValueError("grid does not match the world size")
        self.nprow, self.npcol = nprow, npcol
        self.myrow, self.mycol = divmod(self.rank, npcol)
        self.row_group = self.col_group = None
        # Every process must create every subgroup, in the same order.
        if self.world > 1:
            for r in range(nprow):
                ranks = [r * npcol + c for c in range(npcol)]
                g = dist.new_group(ranks) if npcol > 1 else None
                if r == self.myrow:
                    self.row_group, self.row_ranks = g, ranks
            for c in range(npcol):
                ranks = [r * npcol + c for r in range(nprow)]
                g = dist.new_group(ranks) if nprow > 1 else None
                if c == self.mycol:
                    self.col_group, self.col_ranks = g, ranks
        self._host_groups = None

    def host_groups(self):
        """(row group, column group) whose collectives run on the host only: the groups above when the backend is
        gloo, else gloo groups over the same ranks.  Collective over ALL processes on first use."""
        if self._host_groups is None:
            if self.world == 1 or dist.get_backend() == "gloo":
                self._host_groups = (self.row_group, self.col_group)
            else:
                mine = [None, None]
                for r in range(self.nprow):
                    g = dist.new_group([r * self.npcol + c for c in range(self.npcol)], backend="gloo") if self.npcol > 1 else None
                    if r == self.myrow:
                        mine[0] = g
                for c in range(self.npcol):
                    g = dist.new_group([r * self.npcol + c for r in range(self.nprow)], backend="gloo") if self.nprow > 1 else None
                    if c == self.mycol:
                        mine[1] = g
                self._host_groups = tuple(mine)
        return self._host_groups


class DistMatrix:
    """rows x cols matrix, block nb_r x nb_c, block-cyclic over `grid`; `local` is
    the column-major local part (numroc rows x numroc cols)."""

    def __init__(self, grid, rows, cols, nb_r, nb_c, kind, device="cuda"):
        self.grid, self.rows, self.cols, self.nb_r, self.nb_c, self.kind = grid, rows, cols, nb_r, nb_c, kind
        self.mloc = numroc(rows, nb_r, grid.myrow, grid.nprow)
        self.nloc = numroc(cols, nb_c, grid.mycol, grid.npcol)
        self.local = DeviceMatrix(self.mloc, self.nloc, kind, device=device)

    def global_rows(self):
        return global_indices(self.mloc, self.nb_r, self.grid.myrow, self.grid.nprow)

    def global_cols(self):
        return global_indices(self.nloc, self.nb_c, self.grid.mycol, self.grid.npcol)

    def fill_uniform(self, seed):
        """Local part of the globally defined U[-1,1) matrix (device generator)."""
        g = self.grid
        rc = _lib.lib.eigamd_fill_uniform_cyclic_dev(
            self.kind, ctypes.c_void_p(self.local.ptr), self.mloc, self.nloc, self.local.ld, seed, self.nb_r,
            g.nprow, g.myrow, self.nb_c, g.npcol, g.mycol, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "eigamd_fill_uniform_cyclic_dev")
        return self

    def set_from_global(self, full):
        """Test helper: take this process's entries out of a full numpy matrix."""
        part = full[np.ix_(self.global_rows(), self.global_cols())]
        self.local.as_strided().copy_(torch.from_numpy(np.ascontiguousarray(part)))
        return self


def _flat_real(t):
    return torch.view_as_real(t).reshape(-1) if t.is_complex() else t


def _default_local_gemm(alpha, a, b, beta, c):
    from .device import gemm_dev
    gemm_dev("N", "N", alpha, a, b, beta, c)


def device_broadcast(t, src, group=None):
    """Panel exchange on the communication backend itself (RCCL with backend "nccl"):
    asynchronous, returns the work handle."""
    return dist.broadcast(t, src=src, group=group, async_op=True)


class _Done:
    def wait(self):
        return True


def host_staged_broadcast(t, src, group=None):
    """Panel exchange through a host copy: for backends without a device transport (gloo
    ranks that share one GPU in the tests).  Synchronous."""
    h = t.cpu() if dist.get_rank() == src else torch.empty(t.shape, dtype=t.dtype)
    dist.broadcast(h, src=src, group=group)
    if dist.get_rank() != src:
        t.copy_(h)
    return _Done()


def pgemm(alpha, A, B, beta, C, local_gemm=None, broadcast=None, transport=None, stats=None):
    """C <- alpha*A*B + beta*C (no transposes).  A is M x K with blocks (nb_m, kb), B is
    K x N with blocks (kb, nb_n), C is M x N with blocks (nb_m, nb_n), all on one grid.

    With device operands and no injected `broadcast` the panels travel by one of the transports of
    eigen_amd/transport.py (`transport`: "ipc", "rccl" or None = EIGAMD_SUMMA_TRANSPORT, default auto), all
    panels are queued at once on a copy stream and the compute stream, two receive slots per operand keep the
    exchange one panel ahead of the products, and `stats` (a transport.PanelStats) receives the events of every
    panel.  An injected `broadcast` (the CPU tests under gloo, ranks that share a GPU through a host copy) runs
    the same schedule with one wait per panel on the host."""
    g = C.grid
    if not (A.rows == C.rows and B.cols == C.cols and A.cols == B.rows):
        raise ValueError("shape mismatch")
    if not (A.nb_r == C.nb_r and B.nb_c == C.nb_c and A.nb_c == B.nb_r):
        raise ValueError("block sizes do not conform")
    if local_gemm is None:
        local_gemm = _default_local_gemm
    K, kb = A.cols, A.nb_c
    npanels = (K + kb - 1) // kb
    if npanels == 0:
        if C.mloc and C.nloc:
            local_gemm(alpha, A.local.block(0, 0, A.mloc, 0), B.local.block(0, 0, 0, B.nloc), beta, C.local)
        return C
    dev = C.local.storage.device
    if broadcast is None and dev.type == "cuda" and (g.world > 1 or transport is not None):
        return _pgemm_device(alpha, A, B, beta, C, local_gemm, transport, stats)
    if broadcast is None:
        broadcast = device_broadcast
    tdt = _TORCH_OF_KIND[C.kind]
    # Two receive buffers per operand: panel p+1 travels while panel p is multiplied.
    abuf = [torch.empty(max(C.mloc * kb, 1), dtype=tdt, device=dev) for _ in range(2)] if g.npcol > 1 else None
    bbuf = [torch.empty(max(kb * C.nloc, 1), dtype=tdt, device=dev) for _ in range(2)] if g.nprow > 1 else None

    def issue(p):
        w = min(kb, K - p * kb)
        works = []
        # --- A(:, panel p): owned by grid column p % npcol, local columns (p // npcol)*kb ...
        a_owner, a_off = p % g.npcol, (p // g.npcol) * kb
        if g.npcol == 1:
            a_view = A.local.block(0, a_off, A.mloc, w)
        else:
            if g.mycol == a_owner:
                a_view = A.local.block(0, a_off, A.mloc, w)
                flat = A.local.storage[A.local.offset + a_off * A.local.ld:
                                       A.local.offset + a_off * A.local.ld + A.mloc * w]  # ld == mloc
            else:
                flat = abuf[p % 2][:A.mloc * w]
                a_view = DeviceMatrix(A.mloc, w, A.kind, ld=max(A.mloc, 1), storage=abuf[p % 2])
            if A.mloc * w > 0:
                works.append(broadcast(_flat_real(flat), g.row_ranks[a_owner], g.row_group))
        # --- B(panel p, :): owned by grid row p % nprow, local rows (p // nprow)*kb ...
        b_owner, b_off = p % g.nprow, (p // g.nprow) * kb
        if g.nprow == 1:
            b_view = B.local.block(b_off, 0, w, B.nloc)
        else:
            flat = bbuf[p % 2][:w * B.nloc]
            b_view = DeviceMatrix(w, B.nloc, B.kind, ld=max(w, 1), storage=bbuf[p % 2])
            if g.myrow == b_owner and w * B.nloc > 0:
                # pack the strided row block into the contiguous send buffer
                b_view.as_strided().copy_(B.local.block(b_off, 0, w, B.nloc).as_strided())
            if w * B.nloc > 0:
                works.append(broadcast(_flat_real(flat), g.col_ranks[b_owner], g.col_group))
        return a_view, b_view, works

    cur = issue(0)
    for p in range(npanels):
        nxt = issue(p + 1) if p + 1 < npanels else None
        a_view, b_view, works = cur
        for wk in works:
            wk.wait()
        if C.mloc and C.nloc:
            local_gemm(alpha, a_view, b_view, beta if p == 0 else 1.0, C.local)
        cur = nxt
    return C


_COMM_STREAMS = {}
_MASKED = {}


def _comm_stream(dev):
    key = (os.getpid(), dev.index)
    if key not in _COMM_STREAMS:
        _COMM_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _COMM_STREAMS[key]


def _product_stream(dev):
    """Stream of the local products under the RCCL transport: all compute units but EIGAMD_SUMMA_RCCL_CUS
    (default 32, one per shader engine of every XCD -- the library rounds up to that step; 0: the caller's stream,
    nothing set aside)."""
    n = int(os.environ.get("EIGAMD_SUMMA_RCCL_CUS", "32"))
    if n <= 0:
        return None
    key = (os.getpid(), dev.index, n)
    if key not in _MASKED:
        from .transport import MaskedStream
        _MASKED[key] = MaskedStream(n)
    return _MASKED[key].stream


def _pgemm_device(alpha, A, B, beta, C, local_gemm, transport, stats):
    """The schedule of pgemm on device streams (see pgemm)."""
    from . import transport as tp
    g = C.grid
    K, kb = A.cols, A.nb_c
    npanels = (K + kb - 1) // kb
    dev = C.local.storage.device
    tdt = _TORCH_OF_KIND[C.kind]
    es = C.local.elem_size
    kind = getattr(g, "_transport", None) if transport is None else transport
    if kind is None:
        kind = g._transport = tp.choose(None)
    caller = torch.cuda.current_stream()
    comm = _comm_stream(dev)
    compute = caller
    if kind == "rccl":
        compute = _product_stream(dev) or caller
    abuf = [torch.empty(max(C.mloc * kb, 1), dtype=tdt, device=dev) for _ in range(2)] if g.npcol > 1 else None
    bbuf = [torch.empty(max(kb * C.nloc, 1), dtype=tdt, device=dev) for _ in range(2)] if g.nprow > 1 else None
    nloc_panels_b = (B.mloc + kb - 1) // kb

    peers = wa = wb = bimg = None
    if kind == "ipc":
        # What peers read: A's local part as it lies (the columns of a panel are contiguous, ld == mloc) and an image of
        # B's local part with every row panel contiguous (panel q at element q * kb * nloc, leading dimension = its rows).
        peers = tp.PeerMemory()
        row_host, col_host = g.host_groups()
        if g.npcol > 1:
            wa = tp.Window(A.local.storage, g.row_ranks, row_host, peers)
        if g.nprow > 1:
            bimg = torch.empty(max(B.mloc * B.nloc, 1), dtype=tdt, device=dev)
            for q in range(nloc_panels_b):
                w = min(kb, B.mloc - q * kb)
                if w * B.nloc > 0:
                    DeviceMatrix(w, B.nloc, B.kind, ld=w, storage=bimg, offset=q * kb * B.nloc).as_strided().copy_(
                        B.local.block(q * kb, 0, w, B.nloc).as_strided())
            wb = tp.Window(bimg, g.col_ranks, col_host, peers)
        # every rank's operands are at rest in HBM before anybody reads them
        torch.cuda.synchronize(dev)
        dist.barrier(group=tp.host_group())

    def fetch(p, s):
        """Queue panel p into slot s on the current (copy) stream; returns the views the product reads."""
        w = min(kb, K - p * kb)
        a_owner, a_off = p % g.npcol, (p // g.npcol) * kb
        if g.npcol == 1 or (kind == "ipc" and g.mycol == a_owner):
            a_view = A.local.block(0, a_off, A.mloc, w)
        else:
            a_view = DeviceMatrix(A.mloc, w, A.kind, ld=max(A.mloc, 1), storage=abuf[s])
        if g.npcol > 1 and A.mloc * w > 0:
            if kind == "ipc":
                if g.mycol != a_owner:
                    src = wa.ptr[g.row_ranks[a_owner]] + (A.local.offset + a_off * A.local.ld) * es
                    tp.copy_async(abuf[s].data_ptr(), src, A.mloc * w * es, comm)
            else:
                if g.mycol == a_owner:
                    flat = A.local.storage[A.local.offset + a_off * A.local.ld:
                                           A.local.offset + a_off * A.local.ld + A.mloc * w]
                    a_view = A.local.block(0, a_off, A.mloc, w)
                else:
                    flat = abuf[s][:A.mloc * w]
                dist.broadcast(_flat_real(flat), src=g.row_ranks[a_owner], group=g.row_group, async_op=True).wait()
        b_owner, b_q = p % g.nprow, p // g.nprow
        if g.nprow == 1:
            b_view = B.local.block(b_q * kb, 0, w, B.nloc)
        elif kind == "ipc" and g.myrow == b_owner:
            b_view = DeviceMatrix(w, B.nloc, B.kind, ld=max(w, 1), storage=bimg, offset=b_q * kb * B.nloc)
        else:
            b_view = DeviceMatrix(w, B.nloc, B.kind, ld=max(w, 1), storage=bbuf[s])
        if g.nprow > 1 and w * B.nloc > 0:
            if kind == "ipc":
                if g.myrow != b_owner:
                    src = wb.ptr[g.col_ranks[b_owner]] + b_q * kb * B.nloc * es
                    tp.copy_async(bbuf[s].data_ptr(), src, w * B.nloc * es, comm)
            else:
                if g.myrow == b_owner:
                    b_view.as_strided().copy_(B.local.block(b_q * kb, 0, w, B.nloc).as_strided())
                dist.broadcast(_flat_real(bbuf[s][:w * B.nloc]), src=g.col_ranks[b_owner], group=g.col_group,
                               async_op=True).wait()
        return a_view, b_view

    start = torch.cuda.Event()
    start.record(caller)
    comm.wait_event(start)
    if compute is not caller:
        compute.wait_event(start)
    free = [None, None]
    for p in range(npanels):
        s = p % 2
        ev = stats.events() if stats is not None else [torch.cuda.Event() for _ in range(5)]
        with torch.cuda.stream(comm):
            if free[s] is not None:
                comm.wait_event(free[s])     # the product that read this slot two panels ago
            ev[0].record(comm)
            a_view, b_view = fetch(p, s)
            ev[1].record(comm)
        with torch.cuda.stream(compute):
            ev[2].record(compute)
            compute.wait_event(ev[1])
            ev[3].record(compute)
            if C.mloc and C.nloc:
                local_gemm(alpha, a_view, b_view, beta if p == 0 else 1.0, C.local)
            ev[4].record(compute)
        free[s] = ev[4]
    if compute is not caller:
        caller.wait_event(free[(npanels - 1) % 2])
    if kind == "ipc":
        # Nobody changes an operand (or frees the image of B) while a peer may still be reading it: this rank's own
        # reads have finished once its copy stream is idle, and then everybody says so.
        comm.synchronize()
        dist.barrier(group=tp.host_group())
        peers.close()
    return C
