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
    """The four local steps on the library's device entry points."""

    @staticmethod
    def panel_lu(panel, transp):
        from .lu import getrf_dev
        _, nswaps, first_zero = getrf_dev(panel, max_block_size=16, transp=transp)
        return first_zero, nswaps

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


def pgetrf(A, ops=DeviceOps, broadcast=None):
    """In-place P*A = L*U of a ColCyclicMatrix.  Returns (transpositions: int32 tensor of
    n global row indices on A's device, nb_transpositions, first_zero_pivot)."""
    if broadcast is None:
        broadcast = device_broadcast
    n, nb, P, r = A.n, A.nb, A.world, A.rank
    if nb != lu_block_size(n):
        raise ValueError(f"column block {nb} must equal the LU block size {lu_block_size(n)} for n = {n}")
    dev = A.local.storage.device
    tdt = _TORCH_OF_KIND[A.kind]
    transp = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    pbuf = torch.empty(max(n * nb, 1), dtype=tdt, device=dev)     # the factorised panel, contiguous
    meta = torch.zeros(nb + 2, dtype=torch.int32, device=dev)     # transpositions of the panel, first_zero, nswaps
    first_zero, nswaps = -1, 0
    nblocks = (n + nb - 1) // nb
    for j in range(nblocks):
        k = j * nb
        bs = min(nb, n - k)
        rows_p = n - k
        owner = j % P
        panel = DeviceMatrix(rows_p, bs, A.kind, ld=max(rows_p, 1), storage=pbuf)
        if r == owner:
            loc = (j // P) * nb
            mine = A.local.block(k, loc, rows_p, bs)
            fz, ns = ops.panel_lu(mine, meta[:bs])          # transpositions relative to row k
            panel.as_strided().copy_(mine.as_strided())
            meta[:bs] += k
            meta[nb] = fz
            meta[nb + 1] = ns
        if P > 1:
            works = [broadcast(pbuf[:rows_p * bs], owner), broadcast(meta, owner)]
            for w in works:
                w.wait()
        fz, ns = int(meta[nb].item()), int(meta[nb + 1].item())
        if fz >= 0 and first_zero == -1:
            first_zero = k + fz
        nswaps += ns
        transp[k:k + bs] = meta[:bs]
        # columns this rank owns left / right of the panel
        owned_upto_j = (j - r) // P + 1 if r <= j else 0          # local blocks with global index <= j
        c_right = min(owned_upto_j * nb, A.nloc)
        if r == owner:
            c_left_end = (j // P) * nb
        else:
            c_left_end = c_right
        if c_left_end > 0:
            ops.laswp(A.local.block(0, 0, n, c_left_end), transp, k, k + bs)
        ncr = A.nloc - c_right
        trows = n - k - bs
        if ncr > 0 and trows > 0:
            ops.laswp(A.local.block(0, c_right, n, ncr), transp, k, k + bs)
            a12 = A.local.block(k, c_right, bs, ncr)
            ops.trsm(panel.block(0, 0, bs, bs), a12)
            ops.gemm_minus(panel.block(bs, 0, trows, bs), a12, A.local.block(k + bs, c_right, trows, ncr))
    return transp[:n], nswaps, first_zero
