// Symmetric rank-k update on the device (syrk.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eigamd {

// C <- alpha * A * A^T + beta * C (op = OP_N, A is n x k) or alpha * A^T * A + beta * C
// (op = OP_T, A is k x n) on the `upper` or lower triangle of the n x n matrix C; the other
// triangle is neither read nor written.  Plain transpose for the complex kinds as well (the
// symmetric, not the Hermitian, update).  Asynchronous on `s`.
hipError_t syrk_device(int kind, bool upper, int op, int64_t n, int64_t k, const double alpha[2], const void* A,
                       int64_t lda, const double beta[2], void* C, int64_t ldc, hipStream_t s);

// Hermitian rank-k update (complex kinds): C <- alpha * A * A^H + beta * C (op = OP_N) or
// alpha * A^H * A + beta * C (op = OP_C) on one triangle, alpha and beta real, the imaginary part
// of the diagonal set to zero; nothing is touched when there is no product and beta == 1.
hipError_t herk_device(int kind, bool upper, int op, int64_t n, int64_t k, double alpha, const void* A, int64_t lda,
                       double beta, void* C, int64_t ldc, hipStream_t s);

// Rank-2k update on one triangle of the n x n matrix C: alpha op(A) op(B)^T + alpha op(B) op(A)^T +
// beta C, or with `herm` (complex kinds, beta real) alpha op(A) op(B)^H + conj(alpha) op(B) op(A)^H +
// beta C with the diagonal's imaginary part set to zero.  op = OP_N: A, B are n x k; otherwise k x n.
hipError_t syr2k_device(int kind, bool upper, int op, bool herm, int64_t n, int64_t k, const double alpha[2], const void* A,
                        int64_t lda, const void* B, int64_t ldb, const double beta[2], void* C, int64_t ldc, hipStream_t s);

}  // namespace eigamd
