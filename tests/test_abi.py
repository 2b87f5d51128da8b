"""CPU-side checks of the C-ABI library: it loads, exports every symbol the public
header declares, validates arguments exactly like the reference's BLAS wrapper
(blas/level3_impl.h:47-60), and refuses to compute without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle_lib as ol
import rng_np
import eigen_amd
from eigen_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "eigen_amd_blas.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b([a-z][a-z0-9_]*)\s*\(", text)) - {"defined"})


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert {"sgemm_", "dgemm_", "cgemm_", "zgemm_", "eigamd_gemm_dev", "eigamd_gemm_host"} <= set(names)
    for n in names:
        assert hasattr(_lib.lib, n), n


@pytest.mark.parametrize("kind", range(4))
def test_error_exits_match_the_netlib_table(kind):
    fn = getattr(_lib.lib, ol.SUFFIX[kind] + "gemm_")
    assert ol.run_netlib_chke(kind, fn) == 0


def test_xerbla_name_and_info():
    c = np.zeros((2, 2), order="F")
    ol.reset_xerbla()
    eigen_amd.f77_gemm("N", "N", 2, 2, 2, 1.0, c, 1, c, 2, 0.0, c, 2)  # lda < m
    assert ol.last_xerbla() == (8, "DGEMM ")
    ol.reset_xerbla()
    eigen_amd.f77_gemm("x", "N", 2, 2, 2, 1.0, c, 2, c, 2, 0.0, c, 2)
    assert ol.last_xerbla() == (1, "DGEMM ")
    ol.reset_xerbla()
    z = np.zeros((2, 2), order="F", dtype=np.complex64)
    eigen_amd.f77_gemm("c", "t", 2, 2, 2, 1.0, z, 2, z, 2, 0.0, z, 1)  # lower-case ops are valid; ldc < m
    assert ol.last_xerbla() == (13, "CGEMM ")
    ol.reset_xerbla()


@pytest.mark.parametrize("name", ["sgetrf_", "dgetrf_", "cgetrf_", "zgetrf_"])
def test_getrf_argument_errors(name):
    a = np.zeros(64, dtype=np.complex128)
    ipiv = np.zeros(4, dtype=np.int32)
    i = lambda v: ctypes.byref(ctypes.c_int(v))
    for (m, n, lda, code) in [(-1, 4, 4, 1), (4, -1, 4, 2), (4, 4, 3, 4)]:      # lapack/lu.cpp:16-24
        info = ctypes.c_int(0)
        ol.reset_xerbla()
        getattr(_lib.lib, name)(i(m), i(n), ctypes.c_void_p(a.ctypes.data), i(lda), ctypes.c_void_p(ipiv.ctypes.data),
                                ctypes.byref(info))
        assert info.value == -code and ol.last_xerbla() == (code, name[:-1].upper())
    info = ctypes.c_int(5)
    getattr(_lib.lib, name)(i(0), i(4), ctypes.c_void_p(a.ctypes.data), i(1), ctypes.c_void_p(ipiv.ctypes.data),
                            ctypes.byref(info))                                     # quick return, no device needed
    assert info.value == 0


@pytest.mark.parametrize("name", ["sgetrs_", "dgetrs_", "cgetrs_", "zgetrs_"])
def test_getrs_argument_errors(name):
    a = np.zeros(64, dtype=np.complex128)
    b = np.ones(64, dtype=np.complex128)
    ipiv = np.arange(1, 5, dtype=np.int32)
    i = lambda v: ctypes.byref(ctypes.c_int(v))
    for (trans, n, nrhs, lda, ldb, code) in [(b"X", 4, 2, 4, 4, 1), (b"N", -1, 2, 4, 4, 2), (b"N", 4, -1, 4, 4, 3),
                                             (b"T", 4, 2, 3, 4, 5), (b"C", 4, 2, 4, 3, 8)]:      # lapack/lu.cpp:48-58
        info = ctypes.c_int(0)
        ol.reset_xerbla()
        getattr(_lib.lib, name)(ctypes.c_char_p(trans), i(n), i(nrhs), ctypes.c_void_p(a.ctypes.data), i(lda),
                                ctypes.c_void_p(ipiv.ctypes.data), ctypes.c_void_p(b.ctypes.data), i(ldb), ctypes.byref(info))
        assert info.value == -code and ol.last_xerbla() == (code, name[:-1].upper())
        assert np.all(b == 1)
    ol.reset_xerbla()


def test_quick_returns_need_no_device():
    c = np.full((3, 3), 7.0, order="F")
    a = np.ones((3, 3), order="F")
    eigen_amd.f77_gemm("N", "N", 0, 3, 3, 1.0, a, 3, a, 3, 0.0, c, 3)   # m == 0
    eigen_amd.f77_gemm("N", "N", 3, 0, 3, 1.0, a, 3, a, 3, 0.0, c, 3)   # n == 0
    eigen_amd.f77_gemm("N", "N", 3, 3, 0, 1.0, a, 3, a, 3, 1.0, c, 3)   # k == 0, beta == 1
    assert (c == 7.0).all()


def test_64bit_entry_rejects_bad_arguments():
    x = np.zeros((2, 2), order="F")
    one = np.ones(1)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    rc = _lib.lib.eigamd_gemm_host(_lib.F64, 9, 0, 2, 2, 2, p(one), p(x), 2, p(x), 2, p(one), p(x), 2)
    assert rc == _lib.ERR_ARG
    rc = _lib.lib.eigamd_gemm_host(_lib.F64, 0, 0, 2, 2, 2, p(one), p(x), 2, p(x), 1, p(one), p(x), 2)
    assert rc == _lib.ERR_ARG and b"#10" in _lib.lib.eigamd_last_error()
    rc = _lib.lib.eigamd_gemm_dev(7, 0, 0, 2, 2, 2, p(one), p(x), 2, p(x), 2, p(one), p(x), 2, None)
    assert rc == _lib.ERR_ARG
    # the 64-bit solve entries: side / uplo / op / diag out of range, leading dimensions too small
    assert _lib.lib.eigamd_trsm_host(_lib.F64, 2, 0, 0, 0, 2, 2, p(one), p(x), 2, p(x), 2) == _lib.ERR_ARG
    assert _lib.lib.eigamd_trsm_host(_lib.F64, 0, 0, 3, 0, 2, 2, p(one), p(x), 2, p(x), 2) == _lib.ERR_ARG
    assert _lib.lib.eigamd_trsm_host(_lib.F64, 0, 0, 0, 0, 2, 2, p(one), p(x), 1, p(x), 2) == _lib.ERR_ARG
    assert _lib.lib.eigamd_trsm_host(_lib.F64, 0, 0, 0, 0, 0, 2, p(one), p(x), 1, p(x), 1) == _lib.OK   # m == 0: nothing to do
    assert _lib.lib.eigamd_getrs_dev(_lib.F64, 5, 2, 2, p(x), 2, None, p(x), 2, None) == _lib.ERR_ARG
    assert _lib.lib.eigamd_getrs_dev(_lib.F64, 0, 2, 2, p(x), 1, None, p(x), 2, None) == _lib.ERR_ARG
    assert _lib.lib.eigamd_getrs_dev(_lib.F64, 0, 0, 2, p(x), 1, None, p(x), 1, None) == _lib.OK        # n == 0


@pytest.mark.skipif(_lib.lib.eigamd_device_ok() == 1, reason="a GPU is present")
def test_no_gpu_means_an_error_not_a_cpu_result():
    a = np.ones((4, 4), order="F")
    c = np.zeros((4, 4), order="F")
    with pytest.raises(eigen_amd.EigenAmdError) as e:
        eigen_amd.gemm_host("N", "N", 4, 4, 4, 1.0, a, 4, a, 4, 0.0, c, 4)
    assert e.value.code == _lib.ERR_NO_GPU
    assert (c == 0).all()


@pytest.mark.parametrize("dtype", ol.DTYPES)
def test_host_generator_matches_its_restatement(dtype):
    x = eigen_amd.fill_uniform_host(dtype, 37, 19, seed=12345, ld=41)
    y = rng_np.matrix(dtype, 12345, 37, 19)
    assert np.array_equal(x, y)
    assert np.all(np.abs(x.real) <= 1) and x.real.min() < -0.9 and x.real.max() > 0.9
    assert (x.base[37:, :] == 0).all()  # rows beyond `rows` are not written


def test_tile_shapes():
    for kind in range(4):
        bm, bn, bk = _lib.tile_shape(kind)
        assert bm % 32 == 0 and bn % 32 == 0 and bk * np.dtype(ol.DTYPES[kind]).itemsize in (128, 256)


def test_library_loads_in_a_process_that_imported_torch_first():
    """bench.py and the multi-rank drivers import torch before the library, and a PyTorch wheel ships its own HIP
    runtime, possibly older than the toolkit the library was built with: the library must not require a runtime
    symbol that runtime lacks (hipStreamGetId, version hip_7.1, did once; it is looked up at run time now)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, "-c", "import torch, sys; sys.path.insert(0, %r); import eigen_amd; print('ok')" % root],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "ok" in run.stdout, run.stderr[-2000:]
