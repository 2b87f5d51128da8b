// Products with a symmetric / Hermitian / triangular operand on the device (level3x.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eigamd {

// C <- alpha * A * B + beta * C (left) or alpha * B * A + beta * C (right); A is m x m (left) or
// n x n (right), symmetric -- or Hermitian when `herm` (complex kinds; the imaginary part of
// its diagonal is ignored) -- and only its `upper` / lower triangle is read.
hipError_t symm_device(int kind, bool left, bool upper, bool herm, int64_t m, int64_t n, const double alpha[2],
                       const void* A, int64_t lda, const void* B, int64_t ldb, const double beta[2], void* C, int64_t ldc,
                       hipStream_t s);

// B <- alpha * op(A) * B (left) or alpha * B * op(A) (right); A triangular (`upper` / lower,
// unit diagonal when `unit`), only that triangle is read.
hipError_t trmm_device(int kind, bool left, bool upper, int op, bool unit, int64_t m, int64_t n, const double alpha[2],
                       const void* A, int64_t lda, void* B, int64_t ldb, hipStream_t s);

// C <- alpha * op(A) * op(B) + beta * C with C of the complex kind `ckind` and exactly one of
// A, B holding real scalars of the matching precision (`a_real` / `b_real`).
hipError_t gemm_mixed_device(int ckind, bool a_real, bool b_real, int opa, int opb, int64_t m, int64_t n, int64_t k,
                             const double alpha[2], const void* A, int64_t lda, const void* B, int64_t ldb,
                             const double beta[2], void* C, int64_t ldc, hipStream_t s);

}  // namespace eigamd
