// Thin C entry points over the *reference itself* (Eigen headers under
// /root/reference, compiled in place by oracle/Makefile into oracle/_ref/).
// Used to pin the C oracle, to generate tests/golden/, and as the "reference"
// CPU baseline of bench.py.  TEST INFRASTRUCTURE ONLY; no Eigen source is copied:
// this file only *uses* the public API (operator*, noalias, PartialPivLU).
#include <Eigen/Core>
#include <Eigen/LU>

#include <chrono>
#include <complex>
#include <cstdint>
#include <cstring>
#include <vector>

using namespace Eigen;

namespace {

template <typename S> struct Views {
  typedef Matrix<S, Dynamic, Dynamic, ColMajor> Mat;
  typedef Map<const Mat, 0, OuterStride<> > CMap;
  typedef Map<Mat, 0, OuterStride<> > MMap;
};

// C = beta*C + alpha * op(A) * op(B) through Eigen's expression templates, the way a
// user of the library writes it (dispatch: Eigen/src/Core/products/GeneralMatrixMatrix.h:411-513).
template <typename S>
void opstar(int opa, int opb, int64_t m, int64_t n, int64_t k, const void* palpha, const void* pa, int64_t lda,
            const void* pb, int64_t ldb, const void* pbeta, void* pc, int64_t ldc) {
  typedef typename Views<S>::CMap CMap;
  typedef typename Views<S>::MMap MMap;
  const S alpha = *static_cast<const S*>(palpha), beta = *static_cast<const S*>(pbeta);
  const bool an = (opa == 0 || opa == 3), bn = (opb == 0 || opb == 3);
  CMap A(static_cast<const S*>(pa), an ? m : k, an ? k : m, OuterStride<>(lda));
  CMap B(static_cast<const S*>(pb), bn ? k : n, bn ? n : k, OuterStride<>(ldb));
  MMap C(static_cast<S*>(pc), m, n, OuterStride<>(ldc));
  if (beta == S(0)) C.setZero();
  else if (beta != S(1)) C *= beta;
  if (k == 0) return;
#define OPSTAR_CASE(code, EA, EB) \
  case code: C.noalias() += alpha * ((EA) * (EB)); break;
  switch (opa * 4 + opb) {
    OPSTAR_CASE(0, A, B)
    OPSTAR_CASE(1, A, B.transpose())
    OPSTAR_CASE(2, A, B.adjoint())
    OPSTAR_CASE(3, A, B.conjugate())
    OPSTAR_CASE(4, A.transpose(), B)
    OPSTAR_CASE(5, A.transpose(), B.transpose())
    OPSTAR_CASE(6, A.transpose(), B.adjoint())
    OPSTAR_CASE(7, A.transpose(), B.conjugate())
    OPSTAR_CASE(8, A.adjoint(), B)
    OPSTAR_CASE(9, A.adjoint(), B.transpose())
    OPSTAR_CASE(10, A.adjoint(), B.adjoint())
    OPSTAR_CASE(11, A.adjoint(), B.conjugate())
    OPSTAR_CASE(12, A.conjugate(), B)
    OPSTAR_CASE(13, A.conjugate(), B.transpose())
    OPSTAR_CASE(14, A.conjugate(), B.adjoint())
    OPSTAR_CASE(15, A.conjugate(), B.conjugate())
  }
#undef OPSTAR_CASE
}

// The forms the F77 route cannot express: one real and one complex operand, and a
// destination with an inner stride (test/mixingtypes.cpp, test/product.h:256-269).
// L, R: scalar types of the operands; the result type follows from them.
template <typename L, typename R>
void opstar_ex(int opa, int opb, int64_t m, int64_t n, int64_t k, const void* palpha, const void* pa, int64_t lda,
               const void* pb, int64_t ldb, const void* pbeta, void* pc, int64_t c_inc, int64_t ldc) {
  typedef typename ScalarBinaryOpTraits<L, R>::ReturnType S;
  typedef Map<const Matrix<L, Dynamic, Dynamic>, 0, OuterStride<> > AMap;
  typedef Map<const Matrix<R, Dynamic, Dynamic>, 0, OuterStride<> > BMap;
  typedef Map<Matrix<S, Dynamic, Dynamic>, 0, Stride<Dynamic, Dynamic> > CMapS;
  const S alpha = *static_cast<const S*>(palpha), beta = *static_cast<const S*>(pbeta);
  const bool an = (opa == 0 || opa == 3), bn = (opb == 0 || opb == 3);
  AMap A(static_cast<const L*>(pa), an ? m : k, an ? k : m, OuterStride<>(lda));
  BMap B(static_cast<const R*>(pb), bn ? k : n, bn ? n : k, OuterStride<>(ldb));
  CMapS C(static_cast<S*>(pc), m, n, Stride<Dynamic, Dynamic>(ldc, c_inc));
  if (beta == S(0)) C.setZero();
  else if (beta != S(1)) C *= beta;
  if (k == 0) return;
#define OPSTAR_CASE(code, EA, EB) \
  case code: C.noalias() += alpha * ((EA) * (EB)); break;
  switch (opa * 4 + opb) {
    OPSTAR_CASE(0, A, B)
    OPSTAR_CASE(1, A, B.transpose())
    OPSTAR_CASE(2, A, B.adjoint())
    OPSTAR_CASE(3, A, B.conjugate())
    OPSTAR_CASE(4, A.transpose(), B)
    OPSTAR_CASE(5, A.transpose(), B.transpose())
    OPSTAR_CASE(6, A.transpose(), B.adjoint())
    OPSTAR_CASE(7, A.transpose(), B.conjugate())
    OPSTAR_CASE(8, A.adjoint(), B)
    OPSTAR_CASE(9, A.adjoint(), B.transpose())
    OPSTAR_CASE(10, A.adjoint(), B.adjoint())
    OPSTAR_CASE(11, A.adjoint(), B.conjugate())
    OPSTAR_CASE(12, A.conjugate(), B)
    OPSTAR_CASE(13, A.conjugate(), B.transpose())
    OPSTAR_CASE(14, A.conjugate(), B.adjoint())
    OPSTAR_CASE(15, A.conjugate(), B.conjugate())
  }
#undef OPSTAR_CASE
}

// The timed operation of bench/bench_gemm.cpp:161-165: c.noalias() += a * b in a
// function the optimiser cannot fold into the caller.
template <typename M> __attribute__((noinline)) void gemm_step(const M& a, const M& b, M& c) {
  c.noalias() += a * b;
}

template <typename S> double bench(int64_t m, int64_t n, int64_t k, int tries) {
  typedef Matrix<S, Dynamic, Dynamic> M;
  M a(m, k), b(k, n), c(m, n);
  a.setRandom();
  b.setRandom();
  c.setOnes();  // bench_gemm.cpp:243-246
  gemm_step(a, b, c);  // warm-up (page faults, thread pool)
  double best = 1e300;
  for (int t = 0; t < tries; ++t) {
    auto t0 = std::chrono::steady_clock::now();
    gemm_step(a, b, c);
    auto t1 = std::chrono::steady_clock::now();
    best = std::min(best, std::chrono::duration<double>(t1 - t0).count());
  }
  return best;
}

}  // namespace

extern "C" {

int ref_opstar(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k, const void* alpha, const void* a,
               int64_t lda, const void* b, int64_t ldb, const void* beta, void* c, int64_t ldc) {
  switch (dtype) {
    case 0: opstar<float>(opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc); return 0;
    case 1: opstar<double>(opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc); return 0;
    case 2: opstar<std::complex<float> >(opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc); return 0;
    case 3: opstar<std::complex<double> >(opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc); return 0;
  }
  return -1;
}

// dtype: kind of C.  a_real / b_real: that operand holds real scalars of the same precision.
int ref_opstar_ex(int dtype, int a_real, int b_real, int opa, int opb, int64_t m, int64_t n, int64_t k, const void* alpha,
                  const void* a, int64_t lda, const void* b, int64_t ldb, const void* beta, void* c, int64_t c_inc,
                  int64_t ldc) {
  typedef std::complex<float> cf;
  typedef std::complex<double> cd;
#define EX(L, R) opstar_ex<L, R>(opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, c, c_inc, ldc); return 0
  if (a_real && b_real) return -1;
  switch (dtype) {
    case 0: if (a_real || b_real) return -1; EX(float, float);
    case 1: if (a_real || b_real) return -1; EX(double, double);
    case 2: if (a_real) { EX(float, cf); } else if (b_real) { EX(cf, float); } else { EX(cf, cf); }
    case 3: if (a_real) { EX(double, cd); } else if (b_real) { EX(cd, double); } else { EX(cd, cd); }
  }
#undef EX
  return -1;
}

// Best-of-`tries` wall seconds of one c.noalias() += a*b (multithreaded when built
// with OpenMP: Eigen/src/Core/products/Parallelizer.h:101-176).
double ref_bench_gemm(int dtype, int64_t m, int64_t n, int64_t k, int tries) {
  switch (dtype) {
    case 0: return bench<float>(m, n, k, tries);
    case 1: return bench<double>(m, n, k, tries);
    case 2: return bench<std::complex<float> >(m, n, k, tries);
    case 3: return bench<std::complex<double> >(m, n, k, tries);
  }
  return -1;
}

// partial_lu_impl::blocked_lu itself (Eigen/src/LU/PartialPivLU.h:429-498): in-place
// factors, 0-based row transpositions, number of transpositions; returns the index
// of the first exactly-zero pivot or -1.
long ref_blocked_lu_d(int64_t rows, int64_t cols, double* a, int64_t lda, int* transp, int* nb_transp) {
  int nb = 0;
  Index r = internal::partial_lu_impl<double, ColMajor, int>::blocked_lu(rows, cols, a, lda, transp, nb);
  *nb_transp = nb;
  return (long)r;
}
long ref_blocked_lu_s(int64_t rows, int64_t cols, float* a, int64_t lda, int* transp, int* nb_transp) {
  int nb = 0;
  Index r = internal::partial_lu_impl<float, ColMajor, int>::blocked_lu(rows, cols, a, lda, transp, nb);
  *nb_transp = nb;
  return (long)r;
}

long ref_blocked_lu_c(int64_t rows, int64_t cols, std::complex<float>* a, int64_t lda, int* transp, int* nb_transp) {
  int nb = 0;
  Index r = internal::partial_lu_impl<std::complex<float>, ColMajor, int>::blocked_lu(rows, cols, a, lda, transp, nb);
  *nb_transp = nb;
  return (long)r;
}
long ref_blocked_lu_z(int64_t rows, int64_t cols, std::complex<double>* a, int64_t lda, int* transp, int* nb_transp) {
  int nb = 0;
  Index r = internal::partial_lu_impl<std::complex<double>, ColMajor, int>::blocked_lu(rows, cols, a, lda, transp, nb);
  *nb_transp = nb;
  return (long)r;
}

// What EIGEN_LAPACK_FUNC(getrs) runs (lapack/lu.cpp:62-84) on 0-based transpositions: the same three
// Eigen expressions per value of trans.  op: 0 = N, 1 = T, 2 = C.
}  // extern "C"
namespace {
template <typename S> void getrs_t(int op, int64_t n, int64_t nrhs, S* a, int64_t lda, int* transp0, S* b, int64_t ldb) {
  typename Views<S>::MMap lu(a, n, n, OuterStride<>(lda));
  typename Views<S>::MMap B(b, n, nrhs, OuterStride<>(ldb));
  typedef Map<Transpositions<Dynamic, Dynamic, int> > Pivots;
  if (op == 0) {
    B = Pivots(transp0, n) * B;
    lu.template triangularView<UnitLower>().solveInPlace(B);
    lu.template triangularView<Upper>().solveInPlace(B);
  } else if (op == 1) {
    lu.template triangularView<Upper>().transpose().solveInPlace(B);
    lu.template triangularView<UnitLower>().transpose().solveInPlace(B);
    B = Pivots(transp0, n).transpose() * B;
  } else {
    lu.template triangularView<Upper>().adjoint().solveInPlace(B);
    lu.template triangularView<UnitLower>().adjoint().solveInPlace(B);
    B = Pivots(transp0, n).transpose() * B;
  }
}
}  // namespace
extern "C" {
void ref_getrs(int dtype, int op, int64_t n, int64_t nrhs, void* a, int64_t lda, int* transp0, void* b, int64_t ldb) {
  switch (dtype) {
    case 0: getrs_t<float>(op, n, nrhs, static_cast<float*>(a), lda, transp0, static_cast<float*>(b), ldb); break;
    case 1: getrs_t<double>(op, n, nrhs, static_cast<double*>(a), lda, transp0, static_cast<double*>(b), ldb); break;
    case 2: getrs_t<std::complex<float> >(op, n, nrhs, static_cast<std::complex<float>*>(a), lda, transp0,
                                          static_cast<std::complex<float>*>(b), ldb); break;
    default: getrs_t<std::complex<double> >(op, n, nrhs, static_cast<std::complex<double>*>(a), lda, transp0,
                                            static_cast<std::complex<double>*>(b), ldb); break;
  }
}

double ref_bench_lu(int64_t n, int tries) {
  typedef Matrix<double, Dynamic, Dynamic> M;
  M a = M::Random(n, n);
  double best = 1e300;
  for (int t = 0; t < tries; ++t) {
    auto t0 = std::chrono::steady_clock::now();
    PartialPivLU<M> lu(a);
    auto t1 = std::chrono::steady_clock::now();
    best = std::min(best, std::chrono::duration<double>(t1 - t0).count());
    if (lu.matrixLU()(0, 0) == 12345.678) return -1;
  }
  return best;
}

// Cache sizes used by the blocking heuristic (GeneralBlockPanelKernel.h:2640-2643).
void ref_set_cache_sizes(long l1, long l2, long l3) { Eigen::setCpuCacheSizes(l1, l2, l3); }

// computeProductBlockingSizes for one thread (GeneralBlockPanelKernel.h:340-352).
void ref_blocking_sizes(int dtype, long* k, long* m, long* n) {
  Index kk = *k, mm = *m, nn = *n;
  switch (dtype) {
    case 0: internal::computeProductBlockingSizes<float, float>(kk, mm, nn, Index(1)); break;
    case 1: internal::computeProductBlockingSizes<double, double>(kk, mm, nn, Index(1)); break;
    case 2: internal::computeProductBlockingSizes<std::complex<float>, std::complex<float> >(kk, mm, nn, Index(1)); break;
    default: internal::computeProductBlockingSizes<std::complex<double>, std::complex<double> >(kk, mm, nn, Index(1)); break;
  }
  *k = kk; *m = mm; *n = nn;
}

// Register-tile extents of gebp_traits (GeneralBlockPanelKernel.h:442-455).
void ref_register_tile(int dtype, int* mr, int* nr) {
  switch (dtype) {
    case 0: *mr = internal::gebp_traits<float, float>::mr; *nr = internal::gebp_traits<float, float>::nr; break;
    case 1: *mr = internal::gebp_traits<double, double>::mr; *nr = internal::gebp_traits<double, double>::nr; break;
    case 2: *mr = internal::gebp_traits<std::complex<float>, std::complex<float> >::mr;
            *nr = internal::gebp_traits<std::complex<float>, std::complex<float> >::nr; break;
    default: *mr = internal::gebp_traits<std::complex<double>, std::complex<double> >::mr;
             *nr = internal::gebp_traits<std::complex<double>, std::complex<double> >::nr; break;
  }
}

void ref_set_threads(int n) { Eigen::setNbThreads(n); }
int ref_num_threads(void) { return Eigen::nbThreads(); }
const char* ref_simd(void) { return Eigen::SimdInstructionSetsInUse(); }
int ref_version(void) { return EIGEN_WORLD_VERSION * 10000 + EIGEN_MAJOR_VERSION * 100 + EIGEN_MINOR_VERSION; }

}  // extern "C"
