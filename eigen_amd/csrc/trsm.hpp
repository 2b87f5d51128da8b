// Device triangular solve with many right-hand sides (trsm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eigamd {

// op(A) X = alpha B (left) or X op(A) = alpha B (right); X overwrites the m x n matrix B.
// A is m x m (left) or n x n (right), only its `upper` / lower triangle is read, and not the
// diagonal when `unit`.  op is an Op of gemm_launch.hpp (N, T or C).  Asynchronous on `s`.
hipError_t trsm_device(int kind, bool left, bool upper, int op, bool unit, int64_t m, int64_t n, const double alpha[2],
                       const void* A, int64_t lda, void* B, int64_t ldb, hipStream_t s);

}  // namespace eigamd
