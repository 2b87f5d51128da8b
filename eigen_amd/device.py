"""Device-resident matrices and the device-pointer GEMM entry
(``eigamd_gemm_dev``).  torch is used for HBM allocation and streams only."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import lib

_TORCH_OF_KIND = {_lib.F32: torch.float32, _lib.F64: torch.float64, _lib.C32: torch.complex64,
                  _lib.C64: torch.complex128}
_NP_OF_KIND = {_lib.F32: np.float32, _lib.F64: np.float64, _lib.C32: np.complex64, _lib.C64: np.complex128}
_KIND_OF_TORCH = {v: k for k, v in _TORCH_OF_KIND.items()}
_OPCODE = {"N": _lib.OP_N, "T": _lib.OP_T, "C": _lib.OP_C, "J": _lib.OP_CONJ}


def _stream_ptr(stream=None):
    s = torch.cuda.current_stream() if stream is None else stream
    return ctypes.c_void_p(s.cuda_stream)


class DeviceMatrix:
    """Column-major rows x cols matrix in HBM with leading dimension ``ld``
    (storage: a 1-D torch tensor of ld*cols elements)."""

    def __init__(self, rows, cols, kind, ld=None, device="cuda", storage=None, offset=0):
        self.rows, self.cols, self.kind = int(rows), int(cols), int(kind)
        self.ld = int(max(rows, 1) if ld is None else ld)
        if self.ld < max(self.rows, 1):
            raise ValueError("ld < rows")
        self.offset = int(offset)
        if storage is None:
            storage = torch.empty(self.ld * max(self.cols, 1), dtype=_TORCH_OF_KIND[self.kind], device=device)
        self.storage = storage

    @property
    def elem_size(self):
        return self.storage.element_size()

    @property
    def ptr(self):
        return self.storage.data_ptr() + self.offset * self.elem_size

    def block(self, r0, c0, rows, cols):
        """View of the sub-matrix [r0, r0+rows) x [c0, c0+cols) (same storage)."""
        if r0 < 0 or c0 < 0 or r0 + rows > self.rows or c0 + cols > self.cols:
            raise IndexError("block out of range")
        return DeviceMatrix(rows, cols, self.kind, ld=self.ld, storage=self.storage,
                            offset=self.offset + r0 + c0 * self.ld)

    def fill_uniform(self, seed, stream=None):
        rc = lib.eigamd_fill_uniform_dev(self.kind, ctypes.c_void_p(self.ptr), self.rows, self.cols, self.ld,
                                         seed, _stream_ptr(stream))
        _lib.check(rc, "eigamd_fill_uniform_dev")
        return self

    def as_strided(self):
        """torch view of shape (rows, cols) with column-major strides."""
        return torch.as_strided(self.storage, (self.rows, self.cols), (1, self.ld), self.offset)

    def to_numpy(self):
        return np.asfortranarray(self.as_strided().cpu().numpy())

    @classmethod
    def from_numpy(cls, a, ld=None, device="cuda"):
        a = np.asarray(a)
        kind = {np.dtype(v): k for k, v in _NP_OF_KIND.items()}[a.dtype]
        m = cls(a.shape[0], a.shape[1], kind, ld=ld, device=device)
        m.as_strided().copy_(torch.from_numpy(np.ascontiguousarray(a)))
        return m


def gemm_dev(opa, opb, alpha, a, b, beta, c, stream=None, noalloc=False):
    """c <- alpha * op(a) * op(b) + beta * c on DeviceMatrix operands, asynchronously on
    ``stream`` (default: torch's current stream).  ``noalloc``: through
    ``eigamd_gemm_dev_noalloc`` (nothing is allocated; C of at most 8 MiB)."""
    if not (a.kind == b.kind == c.kind):
        raise ValueError("operands must share one scalar kind")
    oa, ob = _OPCODE[opa], _OPCODE[opb]
    m, n = c.rows, c.cols
    k = a.cols if oa in (_lib.OP_N, _lib.OP_CONJ) else a.rows
    dt = _NP_OF_KIND[c.kind]
    al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
    fn = lib.eigamd_gemm_dev_noalloc if noalloc else lib.eigamd_gemm_dev
    rc = fn(c.kind, oa, ob, m, n, k, ctypes.c_void_p(al.ctypes.data), ctypes.c_void_p(a.ptr), a.ld,
            ctypes.c_void_p(b.ptr), b.ld, ctypes.c_void_p(be.ctypes.data), ctypes.c_void_p(c.ptr),
            c.ld, _stream_ptr(stream))
    _lib.check(rc, "eigamd_gemm_dev_noalloc" if noalloc else "eigamd_gemm_dev")
    return c


def gemm_mixed_dev(opa, opb, alpha, a, b, beta, c, stream=None):
    """c <- alpha * op(a) * op(b) + beta * c with c complex and exactly one of a, b holding real
    scalars of the matching precision (``eigamd_gemm_mixed_dev``)."""
    a_real, b_real = int(a.kind == c.kind - 2), int(b.kind == c.kind - 2)
    oa, ob = _OPCODE[opa], _OPCODE[opb]
    m, n = c.rows, c.cols
    k = a.cols if oa in (_lib.OP_N, _lib.OP_CONJ) else a.rows
    dt = _NP_OF_KIND[c.kind]
    al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
    rc = lib.eigamd_gemm_mixed_dev(c.kind, a_real, b_real, oa, ob, m, n, k, ctypes.c_void_p(al.ctypes.data),
                                   ctypes.c_void_p(a.ptr), a.ld, ctypes.c_void_p(b.ptr), b.ld,
                                   ctypes.c_void_p(be.ctypes.data), ctypes.c_void_p(c.ptr), c.ld, _stream_ptr(stream))
    _lib.check(rc, "eigamd_gemm_mixed_dev")
    return c


def trsm_dev(side, uplo, transa, diag, alpha, a, b, stream=None):
    """op(a) x = alpha b (side 'L') or x op(a) = alpha b (side 'R') on DeviceMatrix operands,
    x overwriting b; asynchronous on ``stream`` (``eigamd_trsm_dev``)."""
    if a.kind != b.kind:
        raise ValueError("operands must share one scalar kind")
    al = np.array([alpha], dtype=_NP_OF_KIND[b.kind])
    rc = lib.eigamd_trsm_dev(b.kind, "LR".index(side.upper()), "UL".index(uplo.upper()), _OPCODE[transa.upper()],
                             "NU".index(diag.upper()), b.rows, b.cols, ctypes.c_void_p(al.ctypes.data),
                             ctypes.c_void_p(a.ptr), a.ld, ctypes.c_void_p(b.ptr), b.ld, _stream_ptr(stream))
    _lib.check(rc, "eigamd_trsm_dev")
    return b


def syrk_dev(uplo, trans, alpha, a, beta, c, stream=None):
    """c <- alpha * a * a^T + beta * c (trans 'N') or alpha * a^T * a + beta * c (trans 'T') on
    the uplo triangle of the square DeviceMatrix c (``eigamd_syrk_dev``)."""
    if a.kind != c.kind:
        raise ValueError("operands must share one scalar kind")
    dt = _NP_OF_KIND[c.kind]
    al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
    k = a.cols if trans.upper() == "N" else a.rows
    rc = lib.eigamd_syrk_dev(c.kind, "UL".index(uplo.upper()), _OPCODE[trans.upper()], c.rows, k,
                             ctypes.c_void_p(al.ctypes.data), ctypes.c_void_p(a.ptr), a.ld,
                             ctypes.c_void_p(be.ctypes.data), ctypes.c_void_p(c.ptr), c.ld, _stream_ptr(stream))
    _lib.check(rc, "eigamd_syrk_dev")
    return c


def herk_dev(uplo, trans, alpha, a, beta, c, stream=None):
    """c <- alpha * a * a^H + beta * c (trans 'N') or alpha * a^H * a + beta * c (trans 'C') on the
    uplo triangle of the square complex DeviceMatrix c; alpha and beta real (``eigamd_herk_dev``)."""
    if a.kind != c.kind:
        raise ValueError("operands must share one scalar kind")
    rdt = np.float32 if c.kind == _lib.C32 else np.float64
    al, be = np.array([alpha], dtype=rdt), np.array([beta], dtype=rdt)
    k = a.cols if trans.upper() == "N" else a.rows
    rc = lib.eigamd_herk_dev(c.kind, "UL".index(uplo.upper()), _OPCODE[trans.upper()], c.rows, k,
                             ctypes.c_void_p(al.ctypes.data), ctypes.c_void_p(a.ptr), a.ld,
                             ctypes.c_void_p(be.ctypes.data), ctypes.c_void_p(c.ptr), c.ld, _stream_ptr(stream))
    _lib.check(rc, "eigamd_herk_dev")
    return c


def syr2k_dev(uplo, trans, alpha, a, b, beta, c, hermitian=False, stream=None):
    """c <- alpha * op(a) * op(b)^T + alpha * op(b) * op(a)^T + beta * c on the uplo triangle of c, or
    the Hermitian form (^H, conj(alpha) in the second term, beta real) (``eigamd_syr2k_dev``)."""
    dt = _NP_OF_KIND[c.kind]
    al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
    k = a.cols if trans.upper() == "N" else a.rows
    rc = lib.eigamd_syr2k_dev(c.kind, "UL".index(uplo.upper()), _OPCODE[trans.upper()], int(bool(hermitian)), c.rows, k,
                              ctypes.c_void_p(al.ctypes.data), ctypes.c_void_p(a.ptr), a.ld, ctypes.c_void_p(b.ptr), b.ld,
                              ctypes.c_void_p(be.ctypes.data), ctypes.c_void_p(c.ptr), c.ld, _stream_ptr(stream))
    _lib.check(rc, "eigamd_syr2k_dev")
    return c


def symm_dev(side, uplo, alpha, a, b, beta, c, hermitian=False, stream=None):
    """c <- alpha * a * b + beta * c (side 'L') or alpha * b * a + beta * c (side 'R'), a symmetric
    (Hermitian if asked) and read through its uplo triangle only (``eigamd_symm_dev``)."""
    dt = _NP_OF_KIND[c.kind]
    al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
    rc = lib.eigamd_symm_dev(c.kind, "LR".index(side.upper()), "UL".index(uplo.upper()), int(bool(hermitian)), c.rows,
                             c.cols, ctypes.c_void_p(al.ctypes.data), ctypes.c_void_p(a.ptr), a.ld, ctypes.c_void_p(b.ptr),
                             b.ld, ctypes.c_void_p(be.ctypes.data), ctypes.c_void_p(c.ptr), c.ld, _stream_ptr(stream))
    _lib.check(rc, "eigamd_symm_dev")
    return c


def trmm_dev(side, uplo, transa, diag, alpha, a, b, stream=None):
    """b <- alpha * op(a) * b (side 'L') or alpha * b * op(a) (side 'R'), a triangular
    (``eigamd_trmm_dev``)."""
    al = np.array([alpha], dtype=_NP_OF_KIND[b.kind])
    rc = lib.eigamd_trmm_dev(b.kind, "LR".index(side.upper()), "UL".index(uplo.upper()), _OPCODE[transa.upper()],
                             "NU".index(diag.upper()), b.rows, b.cols, ctypes.c_void_p(al.ctypes.data),
                             ctypes.c_void_p(a.ptr), a.ld, ctypes.c_void_p(b.ptr), b.ld, _stream_ptr(stream))
    _lib.check(rc, "eigamd_trmm_dev")
    return b
