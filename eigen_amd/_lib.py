"""ctypes binding of libeigen_amd_blas.so (declarations: include/eigen_amd_blas.h).

The library is built in-tree by ``make -C eigen_amd/csrc`` (or
``__graft_entry__.build()``).  There is no fallback: if the shared object is
missing, importing this module raises, and every compute entry point reports
an error when no gfx950 device is visible.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libeigen_amd_blas.so")

F32, F64, C32, C64 = 0, 1, 2, 3
OP_N, OP_T, OP_C, OP_CONJ = 0, 1, 2, 3
OK, ERR_ARG, ERR_DEVICE, ERR_NO_GPU, ERR_NOT_RESIDENT = 0, -1, -2, -3, -4


class LibraryMissing(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `make -C eigen_amd/csrc` "
            "(hipcc --offload-arch=gfx950); eigen_amd has no CPU implementation")
    return ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)


lib = _load()

_i32p = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64

for _name in ("sgemm_", "dgemm_", "cgemm_", "zgemm_"):
    _f = getattr(lib, _name)
    _f.restype = ctypes.c_int
    _f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, _i32p, _i32p, _i32p, _vp, _vp, _i32p, _vp, _i32p,
                   _vp, _vp, _i32p]

lib.eigamd_gemm_host.restype = ctypes.c_int
lib.eigamd_gemm_host.argtypes = [ctypes.c_int] * 3 + [_i64] * 3 + [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64]
lib.eigamd_gemm_dev.restype = ctypes.c_int
lib.eigamd_gemm_dev.argtypes = [ctypes.c_int] * 3 + [_i64] * 3 + [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp]
lib.eigamd_gemm_dev_noalloc.restype = ctypes.c_int
lib.eigamd_gemm_dev_noalloc.argtypes = lib.eigamd_gemm_dev.argtypes
lib.eigamd_gemm_ex_host.restype = ctypes.c_int
lib.eigamd_gemm_ex_host.argtypes = [ctypes.c_int] * 5 + [_i64] * 3 + [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64]
lib.eigamd_gemm_mixed_dev.restype = ctypes.c_int
lib.eigamd_gemm_mixed_dev.argtypes = [ctypes.c_int] * 5 + [_i64] * 3 + [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp]
lib.eigamd_fill_uniform_dev.restype = ctypes.c_int
lib.eigamd_fill_uniform_dev.argtypes = [ctypes.c_int, _vp, _i64, _i64, _i64, ctypes.c_uint64, _vp]
lib.eigamd_fill_uniform_cyclic_dev.restype = ctypes.c_int
lib.eigamd_fill_uniform_cyclic_dev.argtypes = [ctypes.c_int, _vp, _i64, _i64, _i64, ctypes.c_uint64, _i64, ctypes.c_int,
                                               ctypes.c_int, _i64, ctypes.c_int, ctypes.c_int, _vp]
lib.eigamd_fill_uniform_host.restype = ctypes.c_int
lib.eigamd_fill_uniform_host.argtypes = [ctypes.c_int, _vp, _i64, _i64, _i64, ctypes.c_uint64]
_i32ptr = ctypes.POINTER(ctypes.c_int)
lib.eigamd_blocked_lu_host.restype = ctypes.c_int
lib.eigamd_blocked_lu_host.argtypes = [ctypes.c_int, _i64, _i64, _vp, _i64, _vp, _i32ptr, _i32ptr]
lib.eigamd_blocked_lu_host_rowmajor.restype = ctypes.c_int
lib.eigamd_blocked_lu_host_rowmajor.argtypes = lib.eigamd_blocked_lu_host.argtypes
lib.eigamd_getrf_dev.restype = ctypes.c_int
lib.eigamd_getrf_dev.argtypes = [ctypes.c_int, _i64, _i64, _vp, _i64, _vp, _i64, _i32ptr, _i32ptr, _vp]
lib.eigamd_getrf_dev_shared.restype = ctypes.c_int
lib.eigamd_getrf_dev_shared.argtypes = lib.eigamd_getrf_dev.argtypes
lib.eigamd_getrf_dev_async.restype = ctypes.c_int
lib.eigamd_getrf_dev_async.argtypes = [ctypes.c_int, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp]
lib.eigamd_laswp_dev.restype = ctypes.c_int
lib.eigamd_laswp_dev.argtypes = [ctypes.c_int, _vp, _i64, _i64, _vp, ctypes.c_int, ctypes.c_int, _vp]
lib.eigamd_trsm_llnu_dev.restype = ctypes.c_int
lib.eigamd_trsm_llnu_dev.argtypes = [ctypes.c_int, _i64, _i64, _vp, _i64, _vp, _i64, _vp]
for _name in ("dgetrf_", "sgetrf_", "cgetrf_", "zgetrf_"):
    _f = getattr(lib, _name)
    _f.restype = ctypes.c_int
    _f.argtypes = [_i32p, _i32p, _vp, _i32p, _vp, _i32p]
for _name in ("sgetrs_", "dgetrs_", "cgetrs_", "zgetrs_"):
    _f = getattr(lib, _name)
    _f.restype = ctypes.c_int
    _f.argtypes = [ctypes.c_char_p, _i32p, _i32p, _vp, _i32p, _vp, _vp, _i32p, _i32p]
lib.eigamd_getrs_dev.restype = ctypes.c_int
lib.eigamd_getrs_dev.argtypes = [ctypes.c_int, ctypes.c_int, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp]
for _name in ("strsm_", "dtrsm_", "ctrsm_", "ztrsm_"):
    _f = getattr(lib, _name)
    _f.restype = ctypes.c_int
    _f.argtypes = [ctypes.c_char_p] * 4 + [_i32p, _i32p, _vp, _vp, _i32p, _vp, _i32p]
for _name in ("ssyrk_", "dsyrk_", "csyrk_", "zsyrk_"):
    _f = getattr(lib, _name)
    _f.restype = ctypes.c_int
    _f.argtypes = [ctypes.c_char_p] * 2 + [_i32p, _i32p, _vp, _vp, _i32p, _vp, _vp, _i32p]
for _name in ("cherk_", "zherk_"):
    _f = getattr(lib, _name)
    _f.restype = ctypes.c_int
    _f.argtypes = [ctypes.c_char_p] * 2 + [_i32p, _i32p, _vp, _vp, _i32p, _vp, _vp, _i32p]
lib.eigamd_herk_dev.restype = ctypes.c_int
lib.eigamd_herk_dev.argtypes = [ctypes.c_int] * 3 + [_i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp]
for _name in ("ssyr2k_", "dsyr2k_", "csyr2k_", "zsyr2k_", "cher2k_", "zher2k_"):
    _f = getattr(lib, _name)
    _f.restype = ctypes.c_int
    _f.argtypes = [ctypes.c_char_p] * 2 + [_i32p, _i32p, _vp, _vp, _i32p, _vp, _i32p, _vp, _vp, _i32p]
lib.eigamd_syr2k_dev.restype = ctypes.c_int
lib.eigamd_syr2k_dev.argtypes = [ctypes.c_int] * 4 + [_i64, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp]
lib.eigamd_syrk_dev.restype = ctypes.c_int
lib.eigamd_syrk_dev.argtypes = [ctypes.c_int] * 3 + [_i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp]
for _name in ("ssymm_", "dsymm_", "csymm_", "zsymm_", "chemm_", "zhemm_"):
    _f = getattr(lib, _name)
    _f.restype = ctypes.c_int
    _f.argtypes = [ctypes.c_char_p] * 2 + [_i32p, _i32p, _vp, _vp, _i32p, _vp, _i32p, _vp, _vp, _i32p]
for _name in ("strmm_", "dtrmm_", "ctrmm_", "ztrmm_"):
    _f = getattr(lib, _name)
    _f.restype = ctypes.c_int
    _f.argtypes = [ctypes.c_char_p] * 4 + [_i32p, _i32p, _vp, _vp, _i32p, _vp, _i32p]
lib.eigamd_symm_dev.restype = ctypes.c_int
lib.eigamd_symm_dev.argtypes = [ctypes.c_int] * 4 + [_i64, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp]
lib.eigamd_trmm_dev.restype = ctypes.c_int
lib.eigamd_trmm_dev.argtypes = [ctypes.c_int] * 5 + [_i64, _i64, _vp, _vp, _i64, _vp, _i64, _vp]
lib.eigamd_trsm_host.restype = ctypes.c_int
lib.eigamd_trsm_host.argtypes = [ctypes.c_int] * 5 + [_i64, _i64, _vp, _vp, _i64, _vp, _i64]
lib.eigamd_trsm_dev.restype = ctypes.c_int
lib.eigamd_trsm_dev.argtypes = [ctypes.c_int] * 5 + [_i64, _i64, _vp, _vp, _i64, _vp, _i64, _vp]
_i64p = ctypes.POINTER(ctypes.c_int64)
lib.eigamd_ipc_export.restype = ctypes.c_int
lib.eigamd_ipc_export.argtypes = [_vp, _vp, _i64p, _i64p]
lib.eigamd_ipc_open.restype = ctypes.c_int
lib.eigamd_ipc_open.argtypes = [_vp, ctypes.POINTER(ctypes.c_void_p)]
lib.eigamd_ipc_close.restype = ctypes.c_int
lib.eigamd_ipc_close.argtypes = [_vp]
lib.eigamd_copy_dev_async.restype = ctypes.c_int
lib.eigamd_copy_dev_async.argtypes = [_vp, _vp, _i64, _vp]
lib.eigamd_stream_create_cu_mask.restype = ctypes.c_int
lib.eigamd_stream_create_cu_mask.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
lib.eigamd_stream_destroy.restype = ctypes.c_int
lib.eigamd_stream_destroy.argtypes = [_vp]
IPC_HANDLE_BYTES = 64
_pp = ctypes.POINTER(ctypes.c_void_p)
lib.eigamd_pgemm_host.restype = ctypes.c_int
lib.eigamd_pgemm_host.argtypes = [ctypes.c_int] * 3 + [_i64] * 3 + [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, ctypes.c_int,
                                  ctypes.c_int, _i64, _vp, ctypes.c_int]
lib.eigamd_pgemm_create.restype = ctypes.c_int
lib.eigamd_pgemm_create.argtypes = [ctypes.c_int] * 3 + [_i64] * 3 + [ctypes.c_int, ctypes.c_int, _i64, _vp, ctypes.c_int, _pp]
for _name, _args in (("fill", [_vp, ctypes.c_int, ctypes.c_uint64]), ("scatter", [_vp, ctypes.c_int, _vp, _i64]),
                     ("gather", [_vp, ctypes.c_int, _vp, _i64]), ("run", [_vp, _vp, _vp]), ("sync", [_vp]), ("destroy", [_vp]),
                     ("local", [_vp, ctypes.c_int, ctypes.c_int, _pp, ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                ctypes.POINTER(_i64), _i32p])):
    _f = getattr(lib, "eigamd_pgemm_" + _name)
    _f.restype = ctypes.c_int
    _f.argtypes = _args
lib.eigamd_last_error.restype = ctypes.c_char_p
lib.eigamd_last_error.argtypes = []
lib.eigamd_device_ok.restype = ctypes.c_int
lib.eigamd_device_ok.argtypes = []
lib.eigamd_gemm_plan.restype = ctypes.c_int
lib.eigamd_gemm_plan.argtypes = [ctypes.c_int, _i64, _i64, _i64, ctypes.c_int, _i32p]
lib.eigamd_tile_shape.restype = None
lib.eigamd_tile_shape.argtypes = [ctypes.c_int, _i32p, _i32p, _i32p]


class EigenAmdError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__(f"{where}: error {code}: {lib.eigamd_last_error().decode(errors='replace')}")


def check(code, where):
    if code != OK:
        raise EigenAmdError(code, where)


def tile_shape(dtype):
    bm, bn, bk = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.eigamd_tile_shape(dtype, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(bk))
    return bm.value, bn.value, bk.value
