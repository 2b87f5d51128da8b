"""MI355X-native implementation of Eigen's dense matrix-matrix product path.

Python is only the thin host layer used by the tests, the benchmark and the
multi-GPU driver; the product is ``eigen_amd/lib/libeigen_amd_blas.so`` (HIP
kernels + C ABI, ``include/eigen_amd_blas.h``).
"""
from . import _lib
from ._lib import F32, F64, C32, C64, OP_N, OP_T, OP_C, OP_CONJ, EigenAmdError, LibraryMissing
from .blas import gemm, f77_gemm, gemm_host, fill_uniform_host, KIND_OF_DTYPE

__all__ = ["gemm", "f77_gemm", "gemm_host", "fill_uniform_host", "KIND_OF_DTYPE", "F32", "F64", "C32",
           "C64", "OP_N", "OP_T", "OP_C", "OP_CONJ", "EigenAmdError", "LibraryMissing"]
