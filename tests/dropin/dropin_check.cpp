// Drop-in check: an ordinary Eigen program whose dense products (and PartialPivLU) run
// on libeigen_amd_blas.so.  Built twice by oracle/Makefile against the reference
// headers: once with -DEIGEN_USE_BLAS (the reference's own BLAS binding, ?gemm_
// resolved in our library) and once with -DUSE_EIGAMD_SEAM (include/eigen_amd/EigenSeam.h).
// Every product is compared with Eigen's coefficient-based lazyProduct (which never
// reaches a GEMM kernel), in the style of test/product.h and bench/bench_gemm.cpp:256-288.
//
//   dropin_check            -> correctness cases, prints PASS/FAIL lines
//   dropin_check bench N [double|float]
//                           -> c.noalias() += a*b at N^3 through operator* (host pointers): sampled rows
//                              of the result are checked against lazyProduct, then the rate is printed
#include <Eigen/Core>
#include <Eigen/LU>
#ifdef USE_EIGAMD_SEAM
#include <eigen_amd/EigenSeam.h>
#endif

#include <chrono>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace Eigen;

static int g_fail = 0;

template <typename D1, typename D2> static void expect_close(const char* what, const D1& got, const D2& want, double tol) {
  const double denom = want.norm() > 0 ? (double)want.norm() : 1.0;
  const double err = (double)(got - want).norm() / denom;
  const bool ok = err <= tol;
  std::printf("%s %-46s rel_err=%.3e\n", ok ? "PASS" : "FAIL", what, err);
  if (!ok) ++g_fail;
}

template <typename S> static void product_cases(const char* tag, double tol) {
  typedef Matrix<S, Dynamic, Dynamic> Mat;
  typedef Matrix<S, Dynamic, Dynamic, RowMajor> MatR;
  char name[96];
  const int m = 157, n = 93, k = 201;
  Mat a = Mat::Random(m, k), b = Mat::Random(k, n), at = Mat::Random(k, m), bt = Mat::Random(n, k);
  Mat c0 = Mat::Random(m, n);
  const S s = S(2);

  Mat c = a * b;
  std::snprintf(name, sizeof name, "%s C = A*B", tag);
  expect_close(name, c, a.lazyProduct(b).eval(), tol);

  c = c0; c.noalias() += a * b;
  std::snprintf(name, sizeof name, "%s C.noalias() += A*B", tag);
  expect_close(name, c, (c0 + a.lazyProduct(b)).eval(), tol);

  c = c0; c.noalias() -= s * a * b;
  std::snprintf(name, sizeof name, "%s C.noalias() -= s*A*B", tag);
  expect_close(name, c, (c0 - s * a.lazyProduct(b)).eval(), tol);

  c.noalias() = at.transpose() * b;
  std::snprintf(name, sizeof name, "%s C = At^T*B", tag);
  expect_close(name, c, at.transpose().lazyProduct(b).eval(), tol);

  c.noalias() = at.adjoint() * bt.adjoint();
  std::snprintf(name, sizeof name, "%s C = At^H*Bt^H", tag);
  expect_close(name, c, at.adjoint().lazyProduct(bt.adjoint()).eval(), tol);

  c.noalias() = a.conjugate() * bt.transpose();
  std::snprintf(name, sizeof name, "%s C = conj(A)*Bt^T", tag);
  expect_close(name, c, a.conjugate().lazyProduct(bt.transpose()).eval(), tol);

  MatR cr(m, n);
  cr.noalias() = a * b;  // row-major destination: operands swapped, GeneralMatrixMatrix.h:41-46
  std::snprintf(name, sizeof name, "%s RowMajor C = A*B", tag);
  expect_close(name, cr, a.lazyProduct(b).eval(), tol);

  Mat big = Mat::Random(300, 200);
  Mat ref = big;
  big.block(10, 10, 100, 50).noalias() += a.block(0, 0, 100, 64) * b.block(0, 0, 64, 50);
  ref.block(10, 10, 100, 50) += a.block(0, 0, 100, 64).lazyProduct(b.block(0, 0, 64, 50));
  std::snprintf(name, sizeof name, "%s block += blockA*blockB", tag);
  expect_close(name, big, ref, tol);
}

// Triangular solves with many right-hand sides (triangular_solve_matrix): ?trsm_ under EIGEN_USE_BLAS
// (TriangularSolverMatrix_BLAS.h), eigamd_trsm_host under the header seam; and PartialPivLU::solve, which is
// a row permutation and two of them (PartialPivLU.h:227-270; what ?getrs_ runs, lapack/lu.cpp:67-84).
template <typename S> static void solve_cases(const char* tag, double tol) {
  typedef Matrix<S, Dynamic, Dynamic> Mat;
  typedef Matrix<S, Dynamic, Dynamic, RowMajor> MatR;
  char name[96];
  const int n = 230, m = 120;
  // off-diagonal entries small against the diagonal (also against a unit diagonal)
  Mat a = Mat::Random(n, n) * S(4.0 / n);
  a.diagonal() = (a.diagonal().real().array() + 2.0).matrix().template cast<S>();
  const Mat b = Mat::Random(n, m), br = Mat::Random(m, n);

  Mat x = b;
  a.template triangularView<Lower>().solveInPlace(x);
  std::snprintf(name, sizeof name, "%s tri<Lower>.solveInPlace(B)", tag);
  expect_close(name, Mat(a.template triangularView<Lower>()).lazyProduct(x).eval(), b, 50 * tol);

  x = br;
  a.template triangularView<UnitUpper>().template solveInPlace<OnTheRight>(x);
  std::snprintf(name, sizeof name, "%s tri<UnitUpper>.solveInPlace<OnTheRight>(B)", tag);
  expect_close(name, x.lazyProduct(Mat(a.template triangularView<UnitUpper>())).eval(), br, 50 * tol);

  x = b;
  a.adjoint().template triangularView<Lower>().solveInPlace(x);
  std::snprintf(name, sizeof name, "%s A^H tri<Lower>.solveInPlace(B)", tag);
  expect_close(name, Mat(a.adjoint().template triangularView<Lower>()).lazyProduct(x).eval(), b, 50 * tol);

  x = b;
  a.conjugate().template triangularView<Upper>().solveInPlace(x);
  std::snprintf(name, sizeof name, "%s conj(A) tri<Upper>.solveInPlace(B)", tag);
  expect_close(name, Mat(a.conjugate().template triangularView<Upper>()).lazyProduct(x).eval(), b, 50 * tol);

  const MatR ar = a;
  x = b;
  ar.template triangularView<UnitLower>().solveInPlace(x);
  std::snprintf(name, sizeof name, "%s row-major tri<UnitLower>.solveInPlace(B)", tag);
  expect_close(name, Mat(ar.template triangularView<UnitLower>()).lazyProduct(x).eval(), b, 50 * tol);

  const Mat g = Mat::Random(n, n);
  PartialPivLU<Mat> plu(g);
  x = plu.solve(b);
  std::snprintf(name, sizeof name, "%s PartialPivLU.solve(B)", tag);
  expect_close(name, g.lazyProduct(x).eval(), b, 2000 * tol);     // test/lu.cpp:180-183 (m2 ~ m1 * m3)
  x = plu.transpose().solve(b);
  std::snprintf(name, sizeof name, "%s PartialPivLU.transpose().solve(B)", tag);
  expect_close(name, g.transpose().lazyProduct(x).eval(), b, 2000 * tol);
  x = plu.adjoint().solve(b);
  std::snprintf(name, sizeof name, "%s PartialPivLU.adjoint().solve(B)", tag);
  expect_close(name, g.adjoint().lazyProduct(x).eval(), b, 2000 * tol);
}

// Scalar factors, negation, conjugation and transposition folded into one product call by blas_traits
// (Eigen/src/Core/util/BlasUtil.h:404-519): the matrix of test/product_extra.cpp:40-50 (every statement of it,
// with the same operand shapes), then every pairing of {plain, transposed, adjoint, conjugated} left and right
// operands with factors on either side, column- and row-major storage.  The expected value is always formed
// from operands evaluated first, by the coefficient-based lazyProduct, which never reaches a product kernel.
template <typename S, int Order> static void scalar_factor_cases(const char* tag, double tol) {
  typedef Matrix<S, Dynamic, Dynamic, Order> Mat;
  typedef Matrix<S, Dynamic, Dynamic> MatC;
  char name[128];
  const int rows = 83, cols = 61;
  const Mat m1 = Mat::Random(rows, cols), m2 = Mat::Random(rows, cols), square = Mat::Random(rows, rows);
  Mat m3(rows, cols);
  const S s1 = internal::random<S>(), s2 = internal::random<S>(), s3 = internal::random<S>();
  int idx = 0;
#define FOLDED(EXPR, LHS, RHS)                                                                            \
  do {                                                                                                    \
    MatC got_;                                                                                            \
    got_.noalias() = EXPR;                                                                                \
    const MatC l_ = (LHS), r_ = (RHS);                                                                    \
    std::snprintf(name, sizeof name, "%s folded #%d %s", tag, idx++, #EXPR);                              \
    name[90] = 0;                                                                                         \
    expect_close(name, got_, l_.lazyProduct(r_).eval(), tol);                                             \
  } while (0)
  // test/product_extra.cpp:40-50
  FOLDED(m1 * m2.adjoint(), m1, m2.adjoint());
  FOLDED(m1.adjoint() * square.adjoint(), m1.adjoint(), square.adjoint());
  FOLDED(m1.adjoint() * m2, m1.adjoint(), m2);
  FOLDED((s1 * m1.adjoint()) * m2, s1 * m1.adjoint(), m2);
  FOLDED(((s1 * m1).adjoint()) * m2, numext::conj(s1) * m1.adjoint(), m2);
  FOLDED((-m1.adjoint() * s1) * (s3 * m2), -m1.adjoint() * s1, s3 * m2);
  FOLDED((s2 * m1.adjoint() * s1) * m2, s2 * m1.adjoint() * s1, m2);
  FOLDED((-m1 * s2) * s1 * m2.adjoint(), -m1 * s2, s1 * m2.adjoint());
  FOLDED(m1.adjoint() * (s1 * m2).conjugate(), m1.adjoint(), (s1 * m2).conjugate());
  // every pairing of the four forms (inner dimension rows or cols as the forms demand)
  const Mat a = Mat::Random(rows, cols), at = Mat::Random(cols, rows), b = Mat::Random(cols, rows), bt = Mat::Random(rows, cols);
#define LEFT_FORMS(X) X(a) X(at.transpose()) X(at.adjoint()) X(a.conjugate())
#define PAIR_WITH(L)                                                        \
  FOLDED((s1 * L) * (b * s3), s1 * L, b * s3);                              \
  FOLDED((-L) * (s2 * bt.transpose()), -L, s2 * bt.transpose());            \
  FOLDED((L * s2) * (-(bt.adjoint())), L * s2, -(bt.adjoint()));            \
  FOLDED((s3 * L).conjugate().conjugate() * (s1 * b.conjugate()), s3 * L, s1 * b.conjugate());
  LEFT_FORMS(PAIR_WITH)
#undef PAIR_WITH
#undef LEFT_FORMS
  // the same through the accumulating forms
  {
    MatC c0 = MatC::Random(rows, rows), c = c0;
    c.noalias() += (s1 * a) * (s2 * b);
    std::snprintf(name, sizeof name, "%s folded C += (s1*A)*(s2*B)", tag);
    expect_close(name, c, (c0 + MatC(s1 * a).lazyProduct(MatC(s2 * b))).eval(), tol);
    c = c0;
    c.noalias() -= (s3 * at.adjoint()) * bt.adjoint().conjugate().conjugate();
    std::snprintf(name, sizeof name, "%s folded C -= (s3*At^H)*Bt^H", tag);
    expect_close(name, c, (c0 - MatC(s3 * at.adjoint()).lazyProduct(MatC(bt.adjoint()))).eval(), tol);
  }
#undef FOLDED
}

#ifdef EIGEN_USE_BLAS
// The other level-3 routines the reference's headers bind under EIGEN_USE_BLAS
// (TriangularMatrixMatrix_BLAS.h, SelfadjointMatrixMatrix_BLAS.h,
// GeneralMatrixMatrixTriangular_BLAS.h): ?trmm_, ?symm_ / ?hemm_, ?syrk_ resolve in our library.
template <typename S> static void structured_cases(const char* tag, double tol) {
  typedef Matrix<S, Dynamic, Dynamic> Mat;
  char name[96];
  const int n = 230, m = 120;
  // off-diagonal entries small against the diagonal (also against a unit diagonal), real
  // diagonal (a self-adjoint view ignores its imaginary part, a dense copy would not)
  Mat a = Mat::Random(n, n) * S(4.0 / n);
  a.diagonal() = (a.diagonal().real().array() + 2.0).matrix().template cast<S>();
  const Mat b = Mat::Random(n, m), br = Mat::Random(m, n);

  Mat c = a.template triangularView<Upper>() * b;
  std::snprintf(name, sizeof name, "%s C = tri<Upper>(A)*B", tag);
  expect_close(name, c, Mat(a.template triangularView<Upper>()).lazyProduct(b).eval(), tol);

  c = br * a.template triangularView<UnitLower>();
  std::snprintf(name, sizeof name, "%s C = B*tri<UnitLower>(A)", tag);
  expect_close(name, c, br.lazyProduct(Mat(a.template triangularView<UnitLower>())).eval(), tol);

  c = a.template selfadjointView<Lower>() * b;
  std::snprintf(name, sizeof name, "%s C = selfadjoint<Lower>(A)*B", tag);
  expect_close(name, c, Mat(a.template selfadjointView<Lower>()).lazyProduct(b).eval(), tol);

  c = br * a.template selfadjointView<Upper>();
  std::snprintf(name, sizeof name, "%s C = B*selfadjoint<Upper>(A)", tag);
  expect_close(name, c, br.lazyProduct(Mat(a.template selfadjointView<Upper>())).eval(), tol);

  Mat s0 = Mat::Random(n, n), s1 = s0;
  s1.template selfadjointView<Lower>().rankUpdate(b, S(2));
  Mat want = s0;
  want.template triangularView<Lower>() += (S(2) * b.lazyProduct(b.adjoint())).eval();
  std::snprintf(name, sizeof name, "%s selfadjoint<Lower>.rankUpdate(B, 2)", tag);
  expect_close(name, s1, want, tol);
}
#endif

#ifdef USE_EIGAMD_SEAM
// Forms that Eigen's own BLAS binding leaves on the built-in kernel and the header seam binds:
// one real and one complex operand (test/mixingtypes.cpp) and a destination with an inner
// stride (test/product.h:256-269).
template <typename R> static void seam_only_cases(const char* tag, double tol) {
  typedef std::complex<R> Z;
  typedef Matrix<Z, Dynamic, Dynamic> MatZ;
  typedef Matrix<R, Dynamic, Dynamic> MatR;
  char name[96];
  const int m = 157, n = 93, k = 201;
  const MatZ az = MatZ::Random(m, k), bz = MatZ::Random(k, n), azt = MatZ::Random(k, m), c0 = MatZ::Random(m, n);
  const MatR ar = MatR::Random(m, k), br = MatR::Random(k, n), brt = MatR::Random(n, k);
  const MatZ arz = ar.template cast<Z>(), brz = br.template cast<Z>(), brtz = brt.template cast<Z>();
  const Z s(2, -1);

  MatZ c = az * br;
  std::snprintf(name, sizeof name, "%s C = Az*Br (complex x real)", tag);
  expect_close(name, c, az.lazyProduct(brz).eval(), tol);

  c = c0; c.noalias() += ar * bz;
  std::snprintf(name, sizeof name, "%s C += Ar*Bz (real x complex)", tag);
  expect_close(name, c, (c0 + arz.lazyProduct(bz)).eval(), tol);

  c = c0; c.noalias() -= s * (azt.adjoint() * brt.transpose());
  std::snprintf(name, sizeof name, "%s C -= s*Az^H*Br^T", tag);
  expect_close(name, c, (c0 - s * azt.adjoint().lazyProduct(brtz.transpose())).eval(), tol);

  c = c0; c.noalias() += az.conjugate() * br;
  std::snprintf(name, sizeof name, "%s C += conj(Az)*Br", tag);
  expect_close(name, c, (c0 + az.conjugate().lazyProduct(brz)).eval(), tol);

  Matrix<Z, Dynamic, Dynamic, RowMajor> cr(m, n);
  cr.noalias() = ar * bz;
  std::snprintf(name, sizeof name, "%s RowMajor C = Ar*Bz", tag);
  expect_close(name, cr, arz.lazyProduct(bz).eval(), tol);

  // destination with inner stride 2 and outer stride 2*m+3, real and complex
  {
    MatR buf = MatR::Random(2 * m + 3, n), keep = buf;
    Map<MatR, 0, Stride<Dynamic, Dynamic> > cs(buf.data(), m, n, Stride<Dynamic, Dynamic>(2 * m + 3, 2));
    const MatR before = cs;
    cs.noalias() += ar * br;
    std::snprintf(name, sizeof name, "%s strided C += Ar*Br", tag);
    expect_close(name, MatR(cs), (before + ar.lazyProduct(br)).eval(), tol);
    // the elements between the strided ones are untouched
    Map<MatR, 0, Stride<Dynamic, Dynamic> > gaps(buf.data() + 1, m, n, Stride<Dynamic, Dynamic>(2 * m + 3, 2));
    Map<MatR, 0, Stride<Dynamic, Dynamic> > gaps0(keep.data() + 1, m, n, Stride<Dynamic, Dynamic>(2 * m + 3, 2));
    std::snprintf(name, sizeof name, "%s strided C: gaps untouched", tag);
    expect_close(name, MatR(gaps), MatR(gaps0), 0.0);
  }
  {
    MatZ buf = MatZ::Random(3 * m, n);
    Map<MatZ, 0, Stride<Dynamic, Dynamic> > cs(buf.data(), m, n, Stride<Dynamic, Dynamic>(3 * m, 3));
    const MatZ before = cs;
    cs.noalias() -= az * br;
    std::snprintf(name, sizeof name, "%s strided C -= Az*Br", tag);
    expect_close(name, MatZ(cs), (before - az.lazyProduct(brz)).eval(), tol);
  }
}
#endif

static void lu_case(int n) {
  MatrixXd m1 = MatrixXd::Random(n, n);
  PartialPivLU<MatrixXd> plu(m1);
  char name[64];
  std::snprintf(name, sizeof name, "PartialPivLU<MatrixXd>(%d) reconstruct", n);
  expect_close(name, plu.reconstructedMatrix(), m1, 1e-11);  // test/lu.cpp:168
}

template <typename S> static void lu_case_complex(const char* tag, int n, double tol) {
  typedef Matrix<S, Dynamic, Dynamic> Mat;
  typedef Matrix<S, Dynamic, Dynamic, RowMajor> MatR;
  Mat m1 = Mat::Random(n, n);
  PartialPivLU<Mat> plu(m1);
  char name[64];
  std::snprintf(name, sizeof name, "PartialPivLU<%s>(%d) reconstruct", tag, n);
  expect_close(name, plu.reconstructedMatrix(), m1, tol);
  MatR m2 = m1;
  PartialPivLU<MatR> plu2(m2);
  const bool same = plu.permutationP().indices() == plu2.permutationP().indices();
  std::snprintf(name, sizeof name, "%s row-major and column-major LU: same P", tag);
  std::printf("%s %-46s\n", same ? "PASS" : "FAIL", name);
  if (!same) ++g_fail;
}

static void lu_case_rowmajor(int n) {
  typedef Matrix<double, Dynamic, Dynamic, RowMajor> MatR;
  MatR m1 = MatR::Random(n, n);
  PartialPivLU<MatR> plu(m1);
  char name[64];
  std::snprintf(name, sizeof name, "PartialPivLU<RowMajor MatrixXd>(%d) reconstruct", n);
  expect_close(name, plu.reconstructedMatrix(), m1, 1e-11);
  // the same matrix column by column: identical permutation
  MatrixXd m2 = m1;
  PartialPivLU<MatrixXd> plu2(m2);
  const bool same = plu.permutationP().indices() == plu2.permutationP().indices();
  std::printf("%s %-46s\n", same ? "PASS" : "FAIL", "row-major and column-major LU: same permutation");
  if (!same) ++g_fail;
}

template <typename M> __attribute__((noinline)) void gemm_step(const M& a, const M& b, M& c) { c.noalias() += a * b; }

// bench/bench_gemm.cpp:243-246 set-up (A, B random, C ones), three accumulations, then: C must equal
// 1 + 3 A B.  A few rows are recomputed by the coefficient-based product in double precision and
// compared with a normwise tolerance of 8 k eps |a_row| |B| (the bound of north_star, k eps |A| |B|).
template <typename S> static int bench_case(const char* tag, int n) {
  typedef Matrix<S, Dynamic, Dynamic> Mat;
  Mat a = Mat::Random(n, n), b = Mat::Random(n, n), c = Mat::Ones(n, n);
  gemm_step(a, b, c);
  double best = 1e300;
  for (int t = 0; t < 2; ++t) {
    auto t0 = std::chrono::steady_clock::now();
    gemm_step(a, b, c);
    best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  const int rows[5] = {0, 1, n / 3, n / 2 + 1, n - 1};
  const double eps = (double)NumTraits<S>::epsilon();
  int bad = 0;
  for (int r : rows) {
    const RowVectorXd ar = a.row(r).template cast<double>();
    RowVectorXd want(n);
    double bnorm2 = 0;
    for (int j0 = 0; j0 < n; j0 += 512) {  // B in double, 512 columns at a time
      const int nc = std::min(512, n - j0);
      const MatrixXd bj = b.middleCols(j0, nc).template cast<double>();
      want.segment(j0, nc) = ar.lazyProduct(bj);
      if (r == rows[0]) bnorm2 += bj.squaredNorm();
    }
    static double bnorm = 0;
    if (r == rows[0]) bnorm = std::sqrt(bnorm2);
    want = (3.0 * want).array() + 1.0;
    const double err = (c.row(r).template cast<double>() - want).norm();
    const double bound = 8.0 * 3.0 * n * eps * ar.norm() * bnorm / std::sqrt((double)n);
    const bool ok = err <= bound;
    std::printf("%s row %d: |got - want| = %.3e, bound %.3e\n", ok ? "PASS" : "MISMATCH", r, err, bound);
    bad += !ok;
  }
  std::printf("%s operator* %s %d^3: %d sampled rows against lazyProduct\n", bad ? "MISMATCH" : "VERIFIED", tag, n, 5);
  std::printf("operator* %s %d^3 host-to-host: %.4f s  %.2f TFLOP/s\n", tag, n, best, 2.0 * n * n * n / best / 1e12);
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && !std::strcmp(argv[1], "bench")) {
    const int n = std::atoi(argv[2]);
    if (argc >= 4 && !std::strcmp(argv[3], "float")) return bench_case<float>("float", n);
    return bench_case<double>("double", n);
  }
  product_cases<float>("float", 2e-5);
  product_cases<double>("double", 1e-13);
  product_cases<std::complex<float> >("cfloat", 2e-5);
  product_cases<std::complex<double> >("cdouble", 1e-13);
  scalar_factor_cases<float, ColMajor>("float", 2e-5);
  scalar_factor_cases<double, ColMajor>("double", 1e-13);
  scalar_factor_cases<std::complex<float>, ColMajor>("cfloat", 2e-5);
  scalar_factor_cases<std::complex<double>, ColMajor>("cdouble", 1e-13);
  scalar_factor_cases<double, RowMajor>("double rowmajor", 1e-13);
  scalar_factor_cases<std::complex<float>, RowMajor>("cfloat rowmajor", 2e-5);
  scalar_factor_cases<std::complex<double>, RowMajor>("cdouble rowmajor", 1e-13);
  solve_cases<float>("float", 2e-5);
  solve_cases<double>("double", 1e-13);
  solve_cases<std::complex<float> >("cfloat", 2e-5);
  solve_cases<std::complex<double> >("cdouble", 1e-13);
#ifdef EIGEN_USE_BLAS
  structured_cases<float>("float", 2e-5);
  structured_cases<double>("double", 1e-13);
  structured_cases<std::complex<float> >("cfloat", 2e-5);
  structured_cases<std::complex<double> >("cdouble", 1e-13);
#endif
#ifdef USE_EIGAMD_SEAM
  seam_only_cases<float>("float", 2e-5);
  seam_only_cases<double>("double", 1e-13);
#endif
  lu_case(100);
  lu_case(1024);
  lu_case_rowmajor(300);
  lu_case_complex<std::complex<double> >("MatrixXcd", 300, 1e-11);
  lu_case_complex<std::complex<float> >("MatrixXcf", 300, 1e-4);
  std::printf("%s (%d failures)\n", g_fail ? "DROPIN FAIL" : "DROPIN PASS", g_fail);
  return g_fail ? 1 : 0;
}
