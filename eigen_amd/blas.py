"""Host-pointer BLAS surface: the four ``?gemm_`` entry points of Eigen's BLAS
(blas/level3_impl.h:12-76 in the Eigen tree) called on numpy arrays.

Matrices are column-major, as the F77 ABI defines them: pass Fortran-ordered
2-D numpy arrays (or any array whose buffer is the column-major storage, with an
explicit leading dimension).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import lib

KIND_OF_DTYPE = {np.dtype(np.float32): _lib.F32, np.dtype(np.float64): _lib.F64,
                 np.dtype(np.complex64): _lib.C32, np.dtype(np.complex128): _lib.C64}
DTYPE_OF_KIND = {v: k for k, v in KIND_OF_DTYPE.items()}
_F77 = {_lib.F32: lib.sgemm_, _lib.F64: lib.dgemm_, _lib.C32: lib.cgemm_, _lib.C64: lib.zgemm_}
_OPCODE = {"N": _lib.OP_N, "T": _lib.OP_T, "C": _lib.OP_C, "J": _lib.OP_CONJ}


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def f77_gemm(transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    """Raw ``?gemm_`` call: every argument as the F77 ABI takes it.  ``a``, ``b``,
    ``c`` are numpy arrays whose buffers hold the column-major storage; the kind is
    taken from ``c.dtype``.  Invalid arguments go to ``xerbla_`` and the call
    returns without touching ``c``, exactly like the reference."""
    kind = KIND_OF_DTYPE[c.dtype]
    dt = c.dtype
    al = np.array([alpha], dtype=dt)
    be = np.array([beta], dtype=dt)
    ci = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    ta = transa.encode() if isinstance(transa, str) else transa
    tb = transb.encode() if isinstance(transb, str) else transb
    return _F77[kind](ta, tb, ci(m), ci(n), ci(k), _ptr(al), _ptr(a), ci(lda), _ptr(b), ci(ldb), _ptr(be),
                      _ptr(c), ci(ldc))


def _colmajor(x, name):
    if x.ndim != 2 or not x.flags.f_contiguous:
        raise ValueError(f"{name} must be a 2-D Fortran-ordered (column-major) array")
    return x


def gemm(alpha, a, b, beta, c, transa="N", transb="N"):
    """c <- alpha * op(a) * op(b) + beta * c on Fortran-ordered numpy arrays, through
    the F77 entry point of c's dtype."""
    a, b, c = _colmajor(a, "a"), _colmajor(b, "b"), _colmajor(c, "c")
    if not (a.dtype == b.dtype == c.dtype):
        raise ValueError("a, b, c must share one dtype")
    m, n = c.shape
    k = a.shape[1] if transa.upper() == "N" else a.shape[0]
    f77_gemm(transa, transb, m, n, k, alpha, a, max(1, a.shape[0]), b, max(1, b.shape[0]), beta, c,
             max(1, c.shape[0]))
    return c


def gemm_host(opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    """64-bit host entry (``eigamd_gemm_host``): op codes may be 'N','T','C' or 'J'
    (conjugate without transpose).  Raises EigenAmdError on failure."""
    kind = KIND_OF_DTYPE[c.dtype]
    al = np.array([alpha], dtype=c.dtype)
    be = np.array([beta], dtype=c.dtype)
    rc = lib.eigamd_gemm_host(kind, _OPCODE[opa], _OPCODE[opb], m, n, k, _ptr(al), _ptr(a), lda, _ptr(b), ldb,
                              _ptr(be), _ptr(c), ldc)
    _lib.check(rc, "eigamd_gemm_host")
    return c


def gemm_ex_host(opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, c, c_inc, ldc):
    """``eigamd_gemm_ex_host``: the product with a destination whose rows lie ``c_inc``
    elements apart and / or with one operand of real scalars beside a complex one (which
    operand is real is read off the dtypes of ``a`` and ``b``; the kind is that of ``c``)."""
    kind = KIND_OF_DTYPE[c.dtype]
    a_real = int(a.dtype.kind == "f" and c.dtype.kind == "c")
    b_real = int(b.dtype.kind == "f" and c.dtype.kind == "c")
    al = np.array([alpha], dtype=c.dtype)
    be = np.array([beta], dtype=c.dtype)
    rc = lib.eigamd_gemm_ex_host(kind, a_real, b_real, _OPCODE[opa], _OPCODE[opb], m, n, k, _ptr(al), _ptr(a), lda,
                                 _ptr(b), ldb, _ptr(be), _ptr(c), c_inc, ldc)
    _lib.check(rc, "eigamd_gemm_ex_host")
    return c


def fill_uniform_host(dtype, rows, cols, seed, ld=None):
    """Column-major rows x cols matrix of U[-1,1) entries from the library's
    counter-based generator (bit-identical to ``eigamd_fill_uniform_dev``).  Returns a
    Fortran-ordered array; with ``ld`` > rows, a view into an ld x cols buffer."""
    dtype = np.dtype(dtype)
    ld = rows if ld is None else ld
    buf = np.zeros((max(ld, 1), cols), dtype=dtype, order="F")
    rc = lib.eigamd_fill_uniform_host(KIND_OF_DTYPE[dtype], _ptr(buf), rows, cols, max(ld, 1), seed)
    _lib.check(rc, "eigamd_fill_uniform_host")
    return buf[:rows, :]


_F77_TRSM = {_lib.F32: lib.strsm_, _lib.F64: lib.dtrsm_, _lib.C32: lib.ctrsm_, _lib.C64: lib.ztrsm_}


def f77_trsm(side, uplo, transa, diag, m, n, alpha, a, lda, b, ldb):
    """Raw ``?trsm_`` call (blas/level3_impl.h:78-178 in the Eigen tree): op(a) x = alpha b
    (side 'L') or x op(a) = alpha b (side 'R'), x overwriting ``b``; the kind is taken from
    ``b.dtype``."""
    kind = KIND_OF_DTYPE[b.dtype]
    al = np.array([alpha], dtype=b.dtype)
    ci = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    enc = lambda c: c.encode() if isinstance(c, str) else c
    return _F77_TRSM[kind](enc(side), enc(uplo), enc(transa), enc(diag), ci(m), ci(n), _ptr(al), _ptr(a), ci(lda),
                           _ptr(b), ci(ldb))


def trsm(side, uplo, transa, diag, alpha, a, b):
    """Solve in place on Fortran-ordered numpy arrays through the F77 entry point."""
    a, b = _colmajor(a, "a"), _colmajor(b, "b")
    f77_trsm(side, uplo, transa, diag, b.shape[0], b.shape[1], alpha, a, max(a.shape[0], 1), b, max(b.shape[0], 1))
    return b
