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


def pgemm(alpha, A, B, beta, C, local_gemm=None, broadcast=None):
    """C <- alpha*A*B + beta*C (no transposes).  A is M x K with blocks (nb_m, kb), B is
    K x N with blocks (kb, nb_n), C is M x N with blocks (nb_m, nb_n), all on one grid."""
    if broadcast is None:
        broadcast = device_broadcast
    g = C.grid
    if not (A.rows == C.rows and B.cols == C.cols and A.cols == B.rows):
        raise ValueError("shape mismatch")
    if not (A.nb_r == C.nb_r and B.nb_c == C.nb_c and A.nb_c == B.nb_r):
        raise ValueError("block sizes do not conform")
    if local_gemm is None:
        local_gemm = _default_local_gemm
    K, kb = A.cols, A.nb_c
    npanels = (K + kb - 1) // kb
    dev = C.local.storage.device
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

    if npanels == 0:
        if C.mloc and C.nloc:
            local_gemm(alpha, A.local.block(0, 0, A.mloc, 0), B.local.block(0, 0, 0, B.nloc), beta, C.local)
        return C
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
