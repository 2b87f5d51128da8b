// Drop-in check: an ordinary Eigen program whose dense products (and PartialPivLU) run
// on libeigen_amd_blas.so.  Built twice by oracle/Makefile against the reference
// headers: once with -DEIGEN_USE_BLAS (the reference's own BLAS binding, ?gemm_
// resolved in our library) and once with -DUSE_EIGAMD_SEAM (include/eigen_amd/EigenSeam.h).
// Every product is compared with Eigen's coefficient-based lazyProduct (which never
// reaches a GEMM kernel), in the style of test/product.h and bench/bench_gemm.cpp:256-288.
//
//   dropin_check            -> correctness cases, prints PASS/FAIL lines
//   dropin_check bench N    -> times c.noalias() += a*b at N^3 through operator* (host pointers)
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

static void lu_case(int n) {
  MatrixXd m1 = MatrixXd::Random(n, n);
  PartialPivLU<MatrixXd> plu(m1);
  char name[64];
  std::snprintf(name, sizeof name, "PartialPivLU<MatrixXd>(%d) reconstruct", n);
  expect_close(name, plu.reconstructedMatrix(), m1, 1e-11);  // test/lu.cpp:168
}

template <typename M> __attribute__((noinline)) void gemm_step(const M& a, const M& b, M& c) { c.noalias() += a * b; }

int main(int argc, char** argv) {
  if (argc >= 3 && !std::strcmp(argv[1], "bench")) {
    const int n = std::atoi(argv[2]);
    MatrixXd a = MatrixXd::Random(n, n), b = MatrixXd::Random(n, n), c = MatrixXd::Ones(n, n);
    gemm_step(a, b, c);
    double best = 1e300;
    for (int t = 0; t < 2; ++t) {
      auto t0 = std::chrono::steady_clock::now();
      gemm_step(a, b, c);
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    std::printf("operator* double %d^3 host-to-host: %.4f s  %.2f TFLOP/s\n", n, best, 2.0 * n * n * n / best / 1e12);
    return 0;
  }
  product_cases<float>("float", 2e-5);
  product_cases<double>("double", 1e-13);
  product_cases<std::complex<float> >("cfloat", 2e-5);
  product_cases<std::complex<double> >("cdouble", 1e-13);
  lu_case(100);
  lu_case(1024);
  std::printf("%s (%d failures)\n", g_fail ? "DROPIN FAIL" : "DROPIN PASS", g_fail);
  return g_fail ? 1 : 0;
}
