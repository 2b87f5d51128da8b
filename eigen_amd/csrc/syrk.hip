// Symmetric rank-k update C <- alpha * op(A) * op(A)^T + beta * C on one triangle of C.
//
// Replaces general_matrix_matrix_triangular_product + tribb_kernel
// (Eigen/src/Core/products/GeneralMatrixMatrixTriangular.h:62-200) behind
// EIGEN_BLAS_FUNC(syrk) (blas/level3_impl.h:357-433).  The reference runs its GEMM
// micro-kernel on the blocks below (above) the diagonal and a masked variant (through a
// temporary) on the diagonal blocks; here it is ONE launch of the MFMA GEMM over the whole of
// C in which the workgroups of tiles outside the triangle leave at once and the tiles on the
// diagonal write through a row/column predicate in the epilogue, so elements outside the
// triangle are neither read nor written.  beta is fused into those epilogues
// (the reference scales the triangle first, level3_impl.h:396-404: same result).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "gemm_launch.hpp"
#include "syrk.hpp"

namespace eigamd {

namespace {

constexpr int kElem[4] = {4, 8, 8, 16};

hipError_t launch_any(const GemmCall& c, hipStream_t s) {
  switch (c.kind) {
    case F32: return launch_gemm_f32(c, s);
    case F64: return launch_gemm_f64(c, s);
    case C32: return launch_gemm_c32(c, s);
    default:  return launch_gemm_c64(c, s);
  }
}

}  // namespace

hipError_t syrk_device(int kind, bool upper, int op, int64_t n, int64_t k, const double alpha[2], const void* A,
                       int64_t lda, const double beta[2], void* C, int64_t ldc, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int es = kElem[kind];
  const bool notrans = op == OP_N;
  const char* Ab = static_cast<const char*>(A);
  char* Cb = static_cast<char*>(C);
  // rows [r0, r0 + rn) x columns [c0, c0 + cn) of the update
  auto block = [&](int64_t r0, int64_t rn, int64_t c0, int64_t cn, int tri) {
    GemmCall g;
    g.kind = kind;
    g.m = rn; g.n = cn; g.k = k;
    g.alpha[0] = alpha[0]; g.alpha[1] = alpha[1];
    g.beta[0] = beta[0]; g.beta[1] = beta[1];
    if (notrans) {   // A is n x k: rows r0.. times (rows c0..)^T
      g.opa = OP_N; g.A = Ab + (size_t)r0 * es;
      g.opb = OP_T; g.B = Ab + (size_t)c0 * es;
    } else {         // A is k x n: (columns r0..)^T times columns c0..
      g.opa = OP_T; g.A = Ab + (size_t)r0 * lda * es;
      g.opb = OP_N; g.B = Ab + (size_t)c0 * lda * es;
    }
    g.lda = lda; g.ldb = lda;
    g.C = Cb + ((size_t)r0 + (size_t)c0 * ldc) * es;
    g.ldc = ldc;
    g.tri = tri;
    return launch_any(g, s);
  };
  if (k <= 0) {  // only the scaling of the triangle is left (level3_impl.h:396-407)
    if (beta[0] == 1.0 && beta[1] == 0.0) return hipSuccess;
    return block(0, n, 0, n, upper ? 2 : 1);   // k == 0: the launcher runs the scale kernel
  }
  // One launch over the whole of C: workgroups whose tile lies outside the triangle leave at
  // once, the tiles on the diagonal write through the row/column predicate of the epilogue.
  return block(0, n, 0, n, upper ? 2 : 1);
}

namespace {
// Imaginary part of the diagonal <- 0 (what blas/level3_impl.h:613-614, :621 do after the
// Hermitian update).
template <typename Z> __global__ void zero_diag_imag(Z* C, int64_t ldc, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) C[i + (int64_t)i * ldc].y = 0;
}
}  // namespace

hipError_t herk_device(int kind, bool upper, int op, int64_t n, int64_t k, double alpha, const void* A, int64_t lda,
                       double beta, void* C, int64_t ldc, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (kind != C32 && kind != C64) return hipErrorInvalidValue;
  const bool product = k > 0 && alpha != 0.0;                  // level3_impl.h:617
  if (!product && beta == 1.0) return hipSuccess;              // C is not touched at all
  GemmCall g;
  g.kind = kind;
  g.m = n; g.n = n; g.k = product ? k : 0;
  g.alpha[0] = alpha; g.alpha[1] = 0.0;
  g.beta[0] = beta; g.beta[1] = 0.0;
  if (op == OP_N) { g.opa = OP_N; g.opb = OP_C; }              // A (n x k) times A^H
  else            { g.opa = OP_C; g.opb = OP_N; }              // A^H times A (A is k x n)
  g.A = A; g.B = A; g.lda = lda; g.ldb = lda;
  g.C = C; g.ldc = ldc;
  g.tri = upper ? 2 : 1;
  hipError_t e = launch_any(g, s);   // k == 0: the launcher runs the scale kernel on the triangle
  if (e != hipSuccess) return e;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (kind == C32) hipLaunchKernelGGL(zero_diag_imag<float2>, dim3(blocks), dim3(256), 0, s, (float2*)C, ldc, (int)n);
  else hipLaunchKernelGGL(zero_diag_imag<double2>, dim3(blocks), dim3(256), 0, s, (double2*)C, ldc, (int)n);
  return hipGetLastError();
}

// Rank-2k updates (EIGEN_BLAS_FUNC(syr2k) / (her2k), blas/level3_impl.h:435-501, :626-700):
//   C <- alpha op(A) op(B)^T + alpha op(B) op(A)^T + beta C             (symmetric)
//   C <- alpha op(A) op(B)^H + conj(alpha) op(B) op(A)^H + beta C       (Hermitian, beta real)
// on one triangle of C.  The reference writes them as two dense product expressions assigned to
// a triangular view; here they are two launches of the triangular-destination GEMM, the second
// accumulating onto the first.
hipError_t syr2k_device(int kind, bool upper, int op, bool herm, int64_t n, int64_t k, const double alpha[2], const void* A,
                        int64_t lda, const void* B, int64_t ldb, const double beta[2], void* C, int64_t ldc, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (herm && kind != C32 && kind != C64) return hipErrorInvalidValue;
  const bool product = k > 0 && (alpha[0] != 0.0 || alpha[1] != 0.0);
  const bool beta_one = beta[0] == 1.0 && beta[1] == 0.0;
  if (!product && beta_one) return hipSuccess;
  const bool notrans = op == OP_N;
  const int tr = herm ? OP_C : OP_T;
  GemmCall g;
  g.kind = kind;
  g.m = n; g.n = n; g.k = product ? k : 0;
  g.alpha[0] = alpha[0]; g.alpha[1] = alpha[1];
  g.beta[0] = beta[0]; g.beta[1] = herm ? 0.0 : beta[1];
  g.opa = notrans ? OP_N : tr;
  g.opb = notrans ? tr : OP_N;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb;
  g.C = C; g.ldc = ldc;
  g.tri = upper ? 2 : 1;
  hipError_t e = launch_any(g, s);   // k == 0: the scale kernel on the triangle
  if (e != hipSuccess) return e;
  if (product) {
    g.A = B; g.lda = ldb; g.B = A; g.ldb = lda;
    if (herm) g.alpha[1] = -alpha[1];
    g.beta[0] = 1.0; g.beta[1] = 0.0;
    e = launch_any(g, s);
    if (e != hipSuccess) return e;
  }
  if (herm) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (kind == C32) hipLaunchKernelGGL(zero_diag_imag<float2>, dim3(blocks), dim3(256), 0, s, (float2*)C, ldc, (int)n);
    else hipLaunchKernelGGL(zero_diag_imag<double2>, dim3(blocks), dim3(256), 0, s, (double2*)C, ldc, (int)n);
  }
  return hipGetLastError();
}

}  // namespace eigamd
