"""Blocked LU with partial pivoting on the device (eigen_amd/csrc/lu.hip): the host
mirrors of partial_lu_impl::blocked_lu (Eigen/src/LU/PartialPivLU.h:429-498) and of
the LAPACK getrf wrapper (lapack/lu.cpp:14-41)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import lib
from .blas import KIND_OF_DTYPE


def blocked_lu(a):
    """In-place LU of the Fortran-ordered float32/float64/complex64/complex128 array `a` (rows x cols view of a
    column-major buffer).  Returns (row_transpositions, nb_transpositions,
    first_zero_pivot) exactly as the reference's blocked_lu reports them."""
    if a.ndim != 2 or a.strides[0] != a.itemsize:
        raise ValueError("a must be column-major")
    kind = KIND_OF_DTYPE[a.dtype]
    rows, cols = a.shape
    ld = max(1, a.strides[1] // a.itemsize) if cols > 1 else max(1, rows)
    transp = np.zeros(max(min(rows, cols), 1), dtype=np.int32)
    nb, fz = ctypes.c_int(0), ctypes.c_int(-1)
    rc = lib.eigamd_blocked_lu_host(kind, rows, cols, ctypes.c_void_p(a.ctypes.data), ld,
                                    ctypes.c_void_p(transp.ctypes.data), ctypes.byref(nb), ctypes.byref(fz))
    _lib.check(rc, "eigamd_blocked_lu_host")
    return transp[:min(rows, cols)], nb.value, fz.value


def blocked_lu_rowmajor(a):
    """The same on a C-ordered (row-major) array."""
    if a.ndim != 2 or a.strides[1] != a.itemsize:
        raise ValueError("a must be row-major")
    kind = KIND_OF_DTYPE[a.dtype]
    rows, cols = a.shape
    stride = max(1, a.strides[0] // a.itemsize) if rows > 1 else max(1, cols)
    transp = np.zeros(max(min(rows, cols), 1), dtype=np.int32)
    nb, fz = ctypes.c_int(0), ctypes.c_int(-1)
    rc = lib.eigamd_blocked_lu_host_rowmajor(kind, rows, cols, ctypes.c_void_p(a.ctypes.data), stride,
                                             ctypes.c_void_p(transp.ctypes.data), ctypes.byref(nb), ctypes.byref(fz))
    _lib.check(rc, "eigamd_blocked_lu_host_rowmajor")
    return transp[:min(rows, cols)], nb.value, fz.value


def getrf(a):
    """LAPACK-style ?getrf_ on a Fortran-ordered array: returns (ipiv 1-based, info)."""
    kind = KIND_OF_DTYPE[a.dtype]
    fn = {_lib.F32: lib.sgetrf_, _lib.F64: lib.dgetrf_, _lib.C32: lib.cgetrf_, _lib.C64: lib.zgetrf_}[kind]
    m, n = a.shape
    ld = max(1, a.strides[1] // a.itemsize) if n > 1 else max(1, m)
    ipiv = np.zeros(max(min(m, n), 1), dtype=np.int32)
    info = ctypes.c_int(0)
    fn(ctypes.byref(ctypes.c_int(m)), ctypes.byref(ctypes.c_int(n)), ctypes.c_void_p(a.ctypes.data),
       ctypes.byref(ctypes.c_int(ld)), ctypes.c_void_p(ipiv.ctypes.data), ctypes.byref(info))
    return ipiv[:min(m, n)], info.value


class NotResident(RuntimeError):
    """eigamd_getrf_dev gave up because the device is shared (EIGAMD_ERR_NOT_RESIDENT): the matrix is partly
    overwritten; restore it and call getrf_dev(..., shared=True)."""


def getrf_dev(a, stream=None, max_block_size=256, transp=None, shared=False):
    """In-place LU of a DeviceMatrix.  Returns (transpositions as an int32 cuda tensor,
    nb_transpositions, first_zero_pivot).  max_block_size = 16 factorises `a` as one
    panel of the blocked algorithm.  shared=True: eigamd_getrf_dev_shared, the form that needs no
    workgroups resident together (for a device that other work is using)."""
    import torch
    from .device import _stream_ptr
    size = min(a.rows, a.cols)
    if transp is None:
        transp = torch.zeros(max(size, 1), dtype=torch.int32, device=a.storage.device)
    nb, fz = ctypes.c_int(0), ctypes.c_int(-1)
    fn = lib.eigamd_getrf_dev_shared if shared else lib.eigamd_getrf_dev
    rc = fn(a.kind, a.rows, a.cols, ctypes.c_void_p(a.ptr), a.ld, ctypes.c_void_p(transp.data_ptr()),
            max_block_size, ctypes.byref(fz), ctypes.byref(nb), _stream_ptr(stream))
    if rc == _lib.ERR_NOT_RESIDENT:
        raise NotResident(lib.eigamd_last_error().decode())
    _lib.check(rc, "eigamd_getrf_dev")
    return transp[:size], nb.value, fz.value


def getrf_dev_async(a, transp, info, stream=None, max_block_size=256):
    """In-place LU of a DeviceMatrix without waiting for it: `transp` (int32 device tensor,
    min(rows, cols) entries) receives the transpositions and `info` (int32 device tensor, 3
    entries) the first zero pivot or -1, the number of exchanges and a status, all behind the
    stream."""
    from .device import _stream_ptr
    rc = lib.eigamd_getrf_dev_async(a.kind, a.rows, a.cols, ctypes.c_void_p(a.ptr), a.ld, ctypes.c_void_p(transp.data_ptr()),
                                    max_block_size, ctypes.c_void_p(info.data_ptr()), _stream_ptr(stream))
    _lib.check(rc, "eigamd_getrf_dev_async")


def laswp_dev(a, transp, k0, k1, stream=None):
    """Exchange rows i and transp[i] of the DeviceMatrix `a`, i = k0 .. k1-1 in order
    (transp: int32 device tensor of row indices of `a`)."""
    from .device import _stream_ptr
    rc = lib.eigamd_laswp_dev(a.kind, ctypes.c_void_p(a.ptr), a.ld, a.cols, ctypes.c_void_p(transp.data_ptr()), k0, k1,
                              _stream_ptr(stream))
    _lib.check(rc, "eigamd_laswp_dev")


def trsm_llnu_dev(l, b, stream=None):
    """b <- l^-1 b with `l` the unit lower triangle of a square DeviceMatrix block."""
    from .device import _stream_ptr
    rc = lib.eigamd_trsm_llnu_dev(b.kind, b.rows, b.cols, ctypes.c_void_p(l.ptr), l.ld, ctypes.c_void_p(b.ptr), b.ld,
                                  _stream_ptr(stream))
    _lib.check(rc, "eigamd_trsm_llnu_dev")


def getrs(trans, lu, ipiv, b):
    """LAPACK-style ?getrs_ on Fortran-ordered arrays: op(A) X = B from the factors `lu` and the 1-based
    `ipiv` of getrf(); X overwrites `b`.  Returns info."""
    kind = KIND_OF_DTYPE[lu.dtype]
    fn = {_lib.F32: lib.sgetrs_, _lib.F64: lib.dgetrs_, _lib.C32: lib.cgetrs_, _lib.C64: lib.zgetrs_}[kind]
    n = lu.shape[0]
    nrhs = b.shape[1]
    lda = max(1, lu.strides[1] // lu.itemsize) if n > 1 else max(1, n)
    ldb = max(1, b.strides[1] // b.itemsize) if nrhs > 1 else max(1, b.shape[0])
    info = ctypes.c_int(0)
    ci = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    fn(ctypes.c_char_p(trans.encode()), ci(n), ci(nrhs), ctypes.c_void_p(lu.ctypes.data), ci(lda),
       ctypes.c_void_p(ipiv.ctypes.data), ctypes.c_void_p(b.ctypes.data), ci(ldb), ctypes.byref(info))
    return info.value


def getrs_dev(op, lu, transp, b, stream=None):
    """op(A) X = B on DeviceMatrix operands: `lu` and `transp` (int32 device tensor, 0-based) as getrf_dev
    leaves them; X overwrites `b`.  Asynchronous on `stream`."""
    from .device import _stream_ptr
    rc = lib.eigamd_getrs_dev(lu.kind, op, lu.rows, b.cols, ctypes.c_void_p(lu.ptr), lu.ld, ctypes.c_void_p(transp.data_ptr()),
                              ctypes.c_void_p(b.ptr), b.ld, _stream_ptr(stream))
    _lib.check(rc, "eigamd_getrs_dev")
