// Products whose left or right operand is symmetric, Hermitian or triangular:
//   ?symm_ / ?hemm_   (blas/level3_impl.h:245-353, :505-560; product_selfadjoint_matrix,
//                      Eigen/src/Core/products/SelfadjointMatrixMatrix.h)
//   ?trmm_            (blas/level3_impl.h:183-243; product_triangular_matrix_matrix,
//                      Eigen/src/Core/products/TriangularMatrixMatrix.h)
// The reference differs from its GEMM only in the packing stage (symm_pack_lhs/rhs mirror
// the stored triangle while packing, the triangular product packs zero-padded panels), and
// its BLAS wrapper itself goes through dense temporaries for the complex symmetric case
// (:308-322) and for B in trmm (:229-231).  Here the special operand is expanded once into a
// dense matrix in HBM by a small kernel (O(n^2) bytes against O(n^2 m) flop) and the product is
// the MFMA GEMM; for trmm the kernel skips the k-tiles that hold structural zeros only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <mutex>

#include "gemm_launch.hpp"
#include "level3x.hpp"

namespace eigamd {

namespace {

constexpr int kElem[4] = {4, 8, 8, 16};

template <typename T> struct Cx { static constexpr bool value = false; };
template <> struct Cx<float2> { static constexpr bool value = true; };
template <> struct Cx<double2> { static constexpr bool value = true; };

// mode 0: symmetric, 1: Hermitian, 2: triangular (zeros outside), 3: triangular with unit diagonal
template <typename T>
__global__ void expand_special(const T* __restrict__ A, int64_t lda, int n, int upper, int mode, T* __restrict__ X, int64_t ldx) {
  const int64_t total = (int64_t)n * n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(idx / n), i = (int)(idx - (int64_t)j * n);
    const bool stored = upper ? i <= j : i >= j;
    T v = T{};
    if (mode >= 2) {
      if (i == j && mode == 3) {
        if constexpr (Cx<T>::value) v.x = 1; else v = (T)1;
      } else if (stored) {
        v = A[i + (int64_t)j * lda];
      }
    } else {
      v = stored ? A[i + (int64_t)j * lda] : A[j + (int64_t)i * lda];
      if constexpr (Cx<T>::value) {
        if (mode == 1) {
          if (i == j) v.y = 0;
          else if (!stored) v.y = -v.y;
        }
      }
    }
    X[i + (int64_t)j * ldx] = v;
  }
}

template <typename T> __global__ void copy_matrix(const T* __restrict__ S, int64_t lds, int m, int n, T* __restrict__ D, int64_t ldd) {
  const int64_t total = (int64_t)m * n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx / m, i = idx - j * m;
    D[i + j * ldd] = S[i + j * lds];
  }
}

// Real matrix -> complex matrix with zero imaginary parts (the real operand of a mixed
// real x complex product, gebp_traits<complex, real> / <real, complex>,
// GeneralBlockPanelKernel.h:548-681, :925-1049).
template <typename R, typename Z>
__global__ void promote_real(const R* __restrict__ S, int64_t lds, int m, int n, Z* __restrict__ D, int64_t ldd) {
  const int64_t total = (int64_t)m * n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx / m, i = idx - j * m;
    Z z;
    z.x = S[i + j * lds];
    z.y = 0;
    D[i + j * ldd] = z;
  }
}

struct Scratch {
  std::mutex mu;
  void* buf = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;
};
Scratch& scratch() {
  static Scratch s[kMaxDevices];
  return s[current_device_slot()];
}

#define LX_TRY(expr)                 \
  do {                               \
    hipError_t e_ = (expr);          \
    if (e_ != hipSuccess) return e_; \
  } while (0)

// Scratch shared by all callers: a use on stream s waits for the previous use to finish.
hipError_t acquire(Scratch& sc, size_t bytes, hipStream_t s) {
  if (!sc.done) LX_TRY(hipEventCreateWithFlags(&sc.done, hipEventDisableTiming));
  if (bytes > sc.cap) {
    if (sc.buf) {
      LX_TRY(hipEventSynchronize(sc.done));
      LX_TRY(hipFree(sc.buf));
      sc.buf = nullptr;
      sc.cap = 0;
    }
    LX_TRY(hipMalloc(&sc.buf, bytes + bytes / 4 + 4096));
    sc.cap = bytes + bytes / 4 + 4096;
  } else {
    LX_TRY(hipStreamWaitEvent(s, sc.done, 0));
  }
  return hipSuccess;
}

hipError_t expand(int kind, const void* A, int64_t lda, int64_t n, bool upper, int mode, void* X, int64_t ldx, hipStream_t s) {
  const unsigned blocks = (unsigned)std::min<int64_t>(8192, (n * n + 255) / 256);
  switch (kind) {
    case F32: hipLaunchKernelGGL(expand_special<float>, dim3(blocks), dim3(256), 0, s, (const float*)A, lda, (int)n, upper, mode, (float*)X, ldx); break;
    case F64: hipLaunchKernelGGL(expand_special<double>, dim3(blocks), dim3(256), 0, s, (const double*)A, lda, (int)n, upper, mode, (double*)X, ldx); break;
    case C32: hipLaunchKernelGGL(expand_special<float2>, dim3(blocks), dim3(256), 0, s, (const float2*)A, lda, (int)n, upper, mode, (float2*)X, ldx); break;
    default:  hipLaunchKernelGGL(expand_special<double2>, dim3(blocks), dim3(256), 0, s, (const double2*)A, lda, (int)n, upper, mode, (double2*)X, ldx); break;
  }
  return hipGetLastError();
}

hipError_t copy(int kind, const void* S, int64_t lds, int64_t m, int64_t n, void* D, int64_t ldd, hipStream_t s) {
  const unsigned blocks = (unsigned)std::min<int64_t>(8192, (m * n + 255) / 256);
  switch (kind) {
    case F32: hipLaunchKernelGGL(copy_matrix<float>, dim3(blocks), dim3(256), 0, s, (const float*)S, lds, (int)m, (int)n, (float*)D, ldd); break;
    case F64: hipLaunchKernelGGL(copy_matrix<double>, dim3(blocks), dim3(256), 0, s, (const double*)S, lds, (int)m, (int)n, (double*)D, ldd); break;
    case C32: hipLaunchKernelGGL(copy_matrix<float2>, dim3(blocks), dim3(256), 0, s, (const float2*)S, lds, (int)m, (int)n, (float2*)D, ldd); break;
    default:  hipLaunchKernelGGL(copy_matrix<double2>, dim3(blocks), dim3(256), 0, s, (const double2*)S, lds, (int)m, (int)n, (double2*)D, ldd); break;
  }
  return hipGetLastError();
}

hipError_t launch_any(const GemmCall& c, hipStream_t s) {
  switch (c.kind) {
    case F32: return launch_gemm_f32(c, s);
    case F64: return launch_gemm_f64(c, s);
    case C32: return launch_gemm_c32(c, s);
    default:  return launch_gemm_c64(c, s);
  }
}

int64_t round16(int64_t v) { return (v + 15) / 16 * 16; }

}  // namespace

hipError_t symm_device(int kind, bool left, bool upper, bool herm, int64_t m, int64_t n, const double alpha[2],
                       const void* A, int64_t lda, const void* B, int64_t ldb, const double beta[2], void* C, int64_t ldc,
                       hipStream_t s) {
  if (m <= 0 || n <= 0) return hipSuccess;
  const int es = kElem[kind];
  const int64_t na = left ? m : n, ldx = round16(na);
  Scratch& sc = scratch();
  std::lock_guard<std::mutex> lock(sc.mu);
  LX_TRY(acquire(sc, (size_t)ldx * na * es, s));
  LX_TRY(expand(kind, A, lda, na, upper, herm ? 1 : 0, sc.buf, ldx, s));
  GemmCall g;
  g.kind = kind;
  g.opa = OP_N; g.opb = OP_N;
  g.m = m; g.n = n; g.k = na;
  g.alpha[0] = alpha[0]; g.alpha[1] = alpha[1];
  g.beta[0] = beta[0]; g.beta[1] = beta[1];
  if (left) { g.A = sc.buf; g.lda = ldx; g.B = B; g.ldb = ldb; }
  else      { g.A = B; g.lda = ldb; g.B = sc.buf; g.ldb = ldx; }
  g.C = C; g.ldc = ldc;
  LX_TRY(launch_any(g, s));
  return hipEventRecord(sc.done, s);
}

hipError_t trmm_device(int kind, bool left, bool upper, int op, bool unit, int64_t m, int64_t n, const double alpha[2],
                       const void* A, int64_t lda, void* B, int64_t ldb, hipStream_t s) {
  if (m <= 0 || n <= 0) return hipSuccess;
  const int es = kElem[kind];
  const int64_t na = left ? m : n, ldx = round16(na), ldy = round16(m);
  Scratch& sc = scratch();
  std::lock_guard<std::mutex> lock(sc.mu);
  const size_t xbytes = ((size_t)ldx * na * es + 255) / 256 * 256;
  LX_TRY(acquire(sc, xbytes + (size_t)ldy * n * es, s));
  void* X = sc.buf;
  void* Y = static_cast<char*>(sc.buf) + xbytes;
  LX_TRY(expand(kind, A, lda, na, upper, unit ? 3 : 2, X, ldx, s));
  LX_TRY(copy(kind, B, ldb, m, n, Y, ldy, s));   // the product reads the old B while B is rewritten (level3_impl.h:229-231)
  GemmCall g;
  g.kind = kind;
  g.m = m; g.n = n; g.k = na;
  g.alpha[0] = alpha[0]; g.alpha[1] = alpha[1];
  g.beta[0] = 0.0; g.beta[1] = 0.0;
  const bool eff_lower = (!upper) == (op == OP_N);  // op(A) is lower triangular
  if (left) { g.opa = op; g.A = X; g.lda = ldx; g.opb = OP_N; g.B = Y; g.ldb = ldy; g.ktri = eff_lower ? 1 : 2; }
  else      { g.opa = OP_N; g.A = Y; g.lda = ldy; g.opb = op; g.B = X; g.ldb = ldx; g.ktri = eff_lower ? 4 : 3; }
  g.C = B; g.ldc = ldb;
  LX_TRY(launch_any(g, s));
  return hipEventRecord(sc.done, s);
}

// C (complex) <- alpha op(A) op(B) + beta C where exactly one of A, B holds real scalars.
//
// Complex column-major A in plain form times a real B, with real alpha and beta, is a REAL
// product on twice as many rows: the (re, im) pairs of a column of A and of C are consecutive
// real numbers, and a real B acts on both parts alike.  That launch does half the arithmetic
// of a complex product and needs no scratch.  Every other form promotes the real operand to
// complex in scratch memory and runs the complex kernel.
hipError_t gemm_mixed_device(int ckind, bool a_real, bool b_real, int opa, int opb, int64_t m, int64_t n, int64_t k,
                             const double alpha[2], const void* A, int64_t lda, const void* B, int64_t ldb,
                             const double beta[2], void* C, int64_t ldc, hipStream_t s) {
  if (m <= 0 || n <= 0) return hipSuccess;
  if (ckind < C32 || a_real == b_real) return hipErrorInvalidValue;
  const int rkind = ckind - 2;
  // conjugating a real operand does nothing
  auto real_op = [](int op) { return (op == OP_T || op == OP_C) ? OP_T : OP_N; };
  GemmCall g;
  g.m = m; g.n = n; g.k = k;
  g.alpha[0] = alpha[0]; g.alpha[1] = alpha[1];
  g.beta[0] = beta[0]; g.beta[1] = beta[1];
  g.C = C; g.ldc = ldc;
  if (b_real && opa == OP_N && alpha[1] == 0.0 && beta[1] == 0.0 && m < (1 << 30)) {
    g.kind = rkind;
    g.m = 2 * m;
    g.opa = OP_N; g.A = A; g.lda = 2 * lda;
    g.opb = real_op(opb); g.B = B; g.ldb = ldb;
    g.ldc = 2 * ldc;
    return launch_any(g, s);
  }
  if (k <= 0) {  // nothing to promote: C <- beta C
    g.kind = ckind; g.opa = OP_N; g.opb = OP_N; g.A = A; g.lda = std::max<int64_t>(lda, 1); g.B = B; g.ldb = std::max<int64_t>(ldb, 1);
    return launch_any(g, s);
  }
  const int op_r = a_real ? opa : opb;
  const bool plain = (op_r == OP_N || op_r == OP_CONJ);
  const int64_t rr = a_real ? (plain ? m : k) : (plain ? k : n);   // stored rows / columns of the real operand
  const int64_t rc = a_real ? (plain ? k : m) : (plain ? n : k);
  const int64_t ldx = round16(rr);
  Scratch& sc = scratch();
  std::lock_guard<std::mutex> lock(sc.mu);
  LX_TRY(acquire(sc, (size_t)ldx * rc * kElem[ckind], s));
  const unsigned blocks = (unsigned)std::min<int64_t>(8192, (rr * rc + 255) / 256);
  if (ckind == C32)
    hipLaunchKernelGGL((promote_real<float, float2>), dim3(blocks), dim3(256), 0, s, (const float*)(a_real ? A : B),
                       a_real ? lda : ldb, (int)rr, (int)rc, (float2*)sc.buf, ldx);
  else
    hipLaunchKernelGGL((promote_real<double, double2>), dim3(blocks), dim3(256), 0, s, (const double*)(a_real ? A : B),
                       a_real ? lda : ldb, (int)rr, (int)rc, (double2*)sc.buf, ldx);
  LX_TRY(hipGetLastError());
  g.kind = ckind;
  if (a_real) { g.opa = real_op(opa); g.A = sc.buf; g.lda = ldx; g.opb = opb; g.B = B; g.ldb = ldb; }
  else        { g.opa = opa; g.A = A; g.lda = lda; g.opb = real_op(opb); g.B = sc.buf; g.ldb = ldx; }
  LX_TRY(launch_any(g, s));
  return hipEventRecord(sc.done, s);
}

}  // namespace eigamd
