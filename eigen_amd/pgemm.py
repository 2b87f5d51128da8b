"""The product over a grid of devices driven by one process (``eigamd_pgemm_*`` of the C ABI,
eigen_amd/csrc/pgemm.hip): 2-D block-cyclic operands, SUMMA panel exchange one panel ahead of
the local products.  This module is only the ctypes layer."""
import ctypes

import numpy as np

from . import _lib
from ._lib import lib
from .blas import KIND_OF_DTYPE, DTYPE_OF_KIND

_OPCODE = {"N": _lib.OP_N, "T": _lib.OP_T, "C": _lib.OP_C, "J": _lib.OP_CONJ}
PEER, RCCL = 0, 1


def _devarray(devices):
    if devices is None:
        return None
    return (ctypes.c_int * len(devices))(*devices)


def pgemm_host(opa, opb, alpha, a, b, beta, c, nprow, npcol, nb, devices=None, transport=PEER):
    """c <- alpha op(a) op(b) + beta c on Fortran-ordered numpy arrays (views with a larger
    leading dimension allowed), computed on an nprow x npcol grid of devices."""
    kind = KIND_OF_DTYPE[c.dtype]
    m, n = c.shape
    k = a.shape[1] if opa in "NJ" else a.shape[0]
    ld = lambda x: max(1, x.strides[1] // x.itemsize) if x.shape[1] > 1 else max(1, x.shape[0])
    al, be = np.array([alpha], dtype=c.dtype), np.array([beta], dtype=c.dtype)
    rc = lib.eigamd_pgemm_host(kind, _OPCODE[opa], _OPCODE[opb], m, n, k, al.ctypes.data, a.ctypes.data, ld(a), b.ctypes.data,
                               ld(b), be.ctypes.data, c.ctypes.data, ld(c), nprow, npcol, nb, _devarray(devices), transport)
    _lib.check(rc, "eigamd_pgemm_host")
    return c


class Pgemm:
    """Distributed operands kept in HBM between products."""

    def __init__(self, kind, opa, opb, m, n, k, nprow, npcol, nb, devices=None, transport=PEER):
        self.kind, self.m, self.n, self.k = kind, m, n, k
        self.opa, self.opb = opa, opb
        self.nranks = nprow * npcol
        self.h = ctypes.c_void_p()
        rc = lib.eigamd_pgemm_create(kind, _OPCODE[opa], _OPCODE[opb], m, n, k, nprow, npcol, nb, _devarray(devices), transport,
                                     ctypes.byref(self.h))
        _lib.check(rc, "eigamd_pgemm_create")

    def fill(self, which, seed):
        _lib.check(lib.eigamd_pgemm_fill(self.h, "ABC".index(which), seed), "eigamd_pgemm_fill")
        return self

    def stored_shape(self, which):
        if which == "A":
            return (self.m, self.k) if self.opa in "NJ" else (self.k, self.m)
        if which == "B":
            return (self.k, self.n) if self.opb in "NJ" else (self.n, self.k)
        return (self.m, self.n)

    def scatter(self, which, host):
        ld = max(1, host.strides[1] // host.itemsize) if host.shape[1] > 1 else max(1, host.shape[0])
        _lib.check(lib.eigamd_pgemm_scatter(self.h, "ABC".index(which), host.ctypes.data, ld), "eigamd_pgemm_scatter")
        return self

    def gather(self, which):
        rows, cols = self.stored_shape(which)
        out = np.zeros((rows, cols), dtype=DTYPE_OF_KIND[self.kind], order="F")
        _lib.check(lib.eigamd_pgemm_gather(self.h, "ABC".index(which), out.ctypes.data, max(rows, 1)), "eigamd_pgemm_gather")
        return out

    def run(self, alpha, beta):
        dt = DTYPE_OF_KIND[self.kind]
        al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
        _lib.check(lib.eigamd_pgemm_run(self.h, al.ctypes.data, be.ctypes.data), "eigamd_pgemm_run")
        return self

    def sync(self):
        _lib.check(lib.eigamd_pgemm_sync(self.h), "eigamd_pgemm_sync")
        return self

    def close(self):
        if self.h:
            lib.eigamd_pgemm_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
