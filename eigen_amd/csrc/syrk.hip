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

}  // namespace eigamd
