// Header-level seam: routes Eigen's dense product kernel and its blocked
// PartialPivLU to libeigen_amd_blas.so without enabling EIGEN_USE_BLAS /
// EIGEN_USE_LAPACKE (which also reroute gemv, trsm, symm, Cholesky, QR, ...).
//
//   #include <Eigen/Core>
//   #include <Eigen/LU>
//   #include <eigen_amd/EigenSeam.h>     // after the Eigen headers, before any product is instantiated
//
// It adds the two partial specialisations the reference itself adds for external
// back ends -- general_matrix_matrix_product<Index,T,LhsOrder,ConjLhs,T,RhsOrder,ConjRhs,ColMajor,1>
// (Eigen/src/Core/products/GeneralMatrixMatrix_BLAS.h:56-108) and
// partial_lu_impl<T,StorageOrder,int> (Eigen/src/LU/PartialPivLU_LAPACKE.h:44-79) --
// forwarding to the 64-bit C entry points.  Unlike the F77 route, a conjugated
// column-major operand needs no temporary (EIGAMD_OP_CONJ), and the two forms of
// general_matrix_matrix_product that Eigen's BLAS binding leaves on the built-in kernel
// (doc/UsingBlasLapackBackends.dox:59-61) are bound as well: one real and one complex
// operand, and a destination with a non-unit inner stride (ResInnerStride != 1).  The triangular
// solve with many right-hand sides (triangular_solve_matrix, what TriangularView::solveInPlace and
// PartialPivLU::solve run) is bound as Eigen/src/Core/products/TriangularSolverMatrix_BLAS.h binds it.
#ifndef EIGEN_AMD_EIGEN_SEAM_H
#define EIGEN_AMD_EIGEN_SEAM_H

#ifdef EIGEN_USE_BLAS
#error "eigen_amd/EigenSeam.h replaces EIGEN_USE_BLAS for the product path; define only one of them"
#endif

#include <complex>
#include <cstdio>
#include <cstdlib>

#include "../eigen_amd_blas.h"

namespace Eigen {
namespace internal {

template <typename T> struct eigamd_kind;
template <> struct eigamd_kind<float> { enum { value = EIGAMD_F32 }; };
template <> struct eigamd_kind<double> { enum { value = EIGAMD_F64 }; };
template <> struct eigamd_kind<std::complex<float> > { enum { value = EIGAMD_C32 }; };
template <> struct eigamd_kind<std::complex<double> > { enum { value = EIGAMD_C64 }; };

inline void eigamd_require(int rc, const char* what) {
  if (rc != EIGAMD_OK) {
    std::fprintf(stderr, "eigen_amd: %s failed (%d): %s\n", what, rc, eigamd_last_error());
    std::abort();
  }
}

template <typename Index, typename T, int LhsOrder, bool ConjLhs, int RhsOrder, bool ConjRhs>
struct eigamd_product {
  typedef gebp_traits<T, T> Traits;  // read by gemm_functor and parallelize_gemm
  static void run(Index rows, Index cols, Index depth, const T* lhs, Index lhsStride, const T* rhs, Index rhsStride,
                  T* res, Index resIncr, Index resStride, T alpha, level3_blocking<T, T>&, GemmParallelInfo<Index>*) {
    eigen_assert(resIncr == 1);
    EIGEN_ONLY_USED_FOR_DEBUG(resIncr);
    const int opa = LhsOrder == RowMajor ? (ConjLhs ? EIGAMD_OP_C : EIGAMD_OP_T) : (ConjLhs ? EIGAMD_OP_CONJ : EIGAMD_OP_N);
    const int opb = RhsOrder == RowMajor ? (ConjRhs ? EIGAMD_OP_C : EIGAMD_OP_T) : (ConjRhs ? EIGAMD_OP_CONJ : EIGAMD_OP_N);
    const T one(1);
    eigamd_require(eigamd_gemm_host(eigamd_kind<T>::value, opa, opb, rows, cols, depth, &alpha, lhs, lhsStride, rhs,
                                    rhsStride, &one, res, resStride),
                   "eigamd_gemm_host");
  }
};

#define EIGAMD_PRODUCT_SEAM(T)                                                                            \
  template <typename Index, int LhsOrder, bool ConjLhs, int RhsOrder, bool ConjRhs>                       \
  struct general_matrix_matrix_product<Index, T, LhsOrder, ConjLhs, T, RhsOrder, ConjRhs, ColMajor, 1>     \
      : eigamd_product<Index, T, LhsOrder, ConjLhs, RhsOrder, ConjRhs> {};
EIGAMD_PRODUCT_SEAM(float)
EIGAMD_PRODUCT_SEAM(double)
EIGAMD_PRODUCT_SEAM(std::complex<float>)
EIGAMD_PRODUCT_SEAM(std::complex<double>)
#undef EIGAMD_PRODUCT_SEAM

// ResInnerStride other than the constant 1 (a run-time stride, or a fixed stride > 1): rows of
// the destination lie resIncr elements apart (blas_data_mapper with an inner stride,
// GeneralMatrixMatrix.h:73-76).
template <typename Index, typename T, int LhsOrder, bool ConjLhs, int RhsOrder, bool ConjRhs>
struct eigamd_product_strided {
  typedef gebp_traits<T, T> Traits;
  static void run(Index rows, Index cols, Index depth, const T* lhs, Index lhsStride, const T* rhs, Index rhsStride,
                  T* res, Index resIncr, Index resStride, T alpha, level3_blocking<T, T>&, GemmParallelInfo<Index>*) {
    const int opa = LhsOrder == RowMajor ? (ConjLhs ? EIGAMD_OP_C : EIGAMD_OP_T) : (ConjLhs ? EIGAMD_OP_CONJ : EIGAMD_OP_N);
    const int opb = RhsOrder == RowMajor ? (ConjRhs ? EIGAMD_OP_C : EIGAMD_OP_T) : (ConjRhs ? EIGAMD_OP_CONJ : EIGAMD_OP_N);
    const T one(1);
    eigamd_require(eigamd_gemm_ex_host(eigamd_kind<T>::value, 0, 0, opa, opb, rows, cols, depth, &alpha, lhs, lhsStride, rhs,
                                       rhsStride, &one, res, resIncr, resStride),
                   "eigamd_gemm_ex_host");
  }
};
#define EIGAMD_PRODUCT_SEAM_STRIDED(T)                                                                          \
  template <typename Index, int LhsOrder, bool ConjLhs, int RhsOrder, bool ConjRhs, int ResInnerStride>          \
  struct general_matrix_matrix_product<Index, T, LhsOrder, ConjLhs, T, RhsOrder, ConjRhs, ColMajor, ResInnerStride> \
      : eigamd_product_strided<Index, T, LhsOrder, ConjLhs, RhsOrder, ConjRhs> {};
EIGAMD_PRODUCT_SEAM_STRIDED(float)
EIGAMD_PRODUCT_SEAM_STRIDED(double)
EIGAMD_PRODUCT_SEAM_STRIDED(std::complex<float>)
EIGAMD_PRODUCT_SEAM_STRIDED(std::complex<double>)
#undef EIGAMD_PRODUCT_SEAM_STRIDED

// One real and one complex operand (gebp_traits<complex,real> / <real,complex>,
// GeneralBlockPanelKernel.h:548-681, :925-1049); the result and alpha are complex.
template <typename Index, typename L, int LhsOrder, bool ConjLhs, typename R, int RhsOrder, bool ConjRhs>
struct eigamd_product_mixed {
  typedef gebp_traits<L, R> Traits;
  typedef typename ScalarBinaryOpTraits<L, R>::ReturnType ResScalar;
  static void run(Index rows, Index cols, Index depth, const L* lhs, Index lhsStride, const R* rhs, Index rhsStride,
                  ResScalar* res, Index resIncr, Index resStride, ResScalar alpha, level3_blocking<L, R>&,
                  GemmParallelInfo<Index>*) {
    const int opa = LhsOrder == RowMajor ? (ConjLhs ? EIGAMD_OP_C : EIGAMD_OP_T) : (ConjLhs ? EIGAMD_OP_CONJ : EIGAMD_OP_N);
    const int opb = RhsOrder == RowMajor ? (ConjRhs ? EIGAMD_OP_C : EIGAMD_OP_T) : (ConjRhs ? EIGAMD_OP_CONJ : EIGAMD_OP_N);
    const ResScalar one(1);
    eigamd_require(eigamd_gemm_ex_host(eigamd_kind<ResScalar>::value, !NumTraits<L>::IsComplex, !NumTraits<R>::IsComplex, opa,
                                       opb, rows, cols, depth, &alpha, lhs, lhsStride, rhs, rhsStride, &one, res, resIncr,
                                       resStride),
                   "eigamd_gemm_ex_host");
  }
};
#define EIGAMD_PRODUCT_SEAM_MIXED(R)                                                                                   \
  template <typename Index, int LhsOrder, bool ConjLhs, int RhsOrder, bool ConjRhs, int ResInnerStride>                 \
  struct general_matrix_matrix_product<Index, std::complex<R>, LhsOrder, ConjLhs, R, RhsOrder, ConjRhs, ColMajor,        \
                                       ResInnerStride>                                                                  \
      : eigamd_product_mixed<Index, std::complex<R>, LhsOrder, ConjLhs, R, RhsOrder, ConjRhs> {};                        \
  template <typename Index, int LhsOrder, bool ConjLhs, int RhsOrder, bool ConjRhs, int ResInnerStride>                 \
  struct general_matrix_matrix_product<Index, R, LhsOrder, ConjLhs, std::complex<R>, RhsOrder, ConjRhs, ColMajor,        \
                                       ResInnerStride>                                                                  \
      : eigamd_product_mixed<Index, R, LhsOrder, ConjLhs, std::complex<R>, RhsOrder, ConjRhs> {};
EIGAMD_PRODUCT_SEAM_MIXED(float)
EIGAMD_PRODUCT_SEAM_MIXED(double)
#undef EIGAMD_PRODUCT_SEAM_MIXED

// triangular_solve_matrix with a column-major, unit-inner-stride right-hand side, as
// Eigen/src/Core/products/TriangularSolverMatrix_BLAS.h:43-101 (left) and :104-160 (right) bind it:
// a row-major triangle is the transposed column-major one (other triangle, op T, or C when
// conjugated); a conjugated column-major triangle goes through a conjugated copy, as there.
// TriangularView::solveInPlace and PartialPivLU::solve reach the device through this.
template <typename T, typename Index, int Side, int Mode, bool Conjugate, int TriStorageOrder>
struct eigamd_trsolve {
  static void run(Index size, Index otherSize, const T* tri, Index triStride, T* other, Index otherIncr, Index otherStride,
                  level3_blocking<T, T>&) {
    eigen_assert(otherIncr == 1);
    EIGEN_ONLY_USED_FOR_DEBUG(otherIncr);
    const bool lower = (Mode & Lower) == Lower;
    const bool row_major = TriStorageOrder == RowMajor;
    const int uplo = (lower != row_major) ? 1 /* EIGAMD_UPLO_LOWER */ : 0 /* EIGAMD_UPLO_UPPER */;
    const int op = row_major ? (Conjugate ? EIGAMD_OP_C : EIGAMD_OP_T) : EIGAMD_OP_N;
    const int diag = (Mode & UnitDiag) ? 1 : 0;
    const int side = Side == OnTheLeft ? 0 : 1;
    const Index m = Side == OnTheLeft ? size : otherSize, n = Side == OnTheLeft ? otherSize : size;
    typedef Matrix<T, Dynamic, Dynamic, ColMajor> MatrixTri;
    MatrixTri conj_copy;
    const T* a = tri;
    Index lda = triStride;
    if (!row_major && Conjugate && NumTraits<T>::IsComplex) {
      conj_copy = Map<const MatrixTri, 0, OuterStride<> >(tri, size, size, OuterStride<>(triStride)).conjugate();
      a = conj_copy.data();
      lda = conj_copy.outerStride();
    }
    const T one(1);
    eigamd_require(eigamd_trsm_host(eigamd_kind<T>::value, side, uplo, op, diag, m, n, &one, a, lda, other, otherStride),
                   "eigamd_trsm_host");
  }
};
#define EIGAMD_TRSOLVE_SEAM(T)                                                                        \
  template <typename Index, int Mode, bool Conjugate, int TriStorageOrder>                            \
  struct triangular_solve_matrix<T, Index, OnTheLeft, Mode, Conjugate, TriStorageOrder, ColMajor, 1>   \
      : eigamd_trsolve<T, Index, OnTheLeft, Mode, Conjugate, TriStorageOrder> {};                      \
  template <typename Index, int Mode, bool Conjugate, int TriStorageOrder>                            \
  struct triangular_solve_matrix<T, Index, OnTheRight, Mode, Conjugate, TriStorageOrder, ColMajor, 1>  \
      : eigamd_trsolve<T, Index, OnTheRight, Mode, Conjugate, TriStorageOrder> {};
EIGAMD_TRSOLVE_SEAM(float)
EIGAMD_TRSOLVE_SEAM(double)
EIGAMD_TRSOLVE_SEAM(std::complex<float>)
EIGAMD_TRSOLVE_SEAM(std::complex<double>)
#undef EIGAMD_TRSOLVE_SEAM

#ifdef EIGEN_PARTIALLU_H
template <typename T> struct eigamd_lu {
  static Index blocked_lu(Index rows, Index cols, T* lu_data, Index luStride, int* row_transpositions,
                          int& nb_transpositions, Index /*maxBlockSize*/ = 256) {
    int first_zero = -1, nb = 0;
    eigamd_require(eigamd_blocked_lu_host(eigamd_kind<T>::value, rows, cols, lu_data, luStride, row_transpositions, &nb,
                                          &first_zero),
                   "eigamd_blocked_lu_host");
    nb_transpositions = nb;
    return first_zero;
  }
};
template <typename T> struct eigamd_lu_rowmajor {
  static Index blocked_lu(Index rows, Index cols, T* lu_data, Index luStride, int* row_transpositions,
                          int& nb_transpositions, Index /*maxBlockSize*/ = 256) {
    int first_zero = -1, nb = 0;
    eigamd_require(eigamd_blocked_lu_host_rowmajor(eigamd_kind<T>::value, rows, cols, lu_data, luStride, row_transpositions, &nb,
                                                   &first_zero),
                   "eigamd_blocked_lu_host_rowmajor");
    nb_transpositions = nb;
    return first_zero;
  }
};
// Both storage orders and the four scalar kinds, as Eigen/src/LU/PartialPivLU_LAPACKE.h:44-79 binds them.
template <> struct partial_lu_impl<double, ColMajor, int> : eigamd_lu<double> {};
template <> struct partial_lu_impl<float, ColMajor, int> : eigamd_lu<float> {};
template <> struct partial_lu_impl<double, RowMajor, int> : eigamd_lu_rowmajor<double> {};
template <> struct partial_lu_impl<float, RowMajor, int> : eigamd_lu_rowmajor<float> {};
template <> struct partial_lu_impl<std::complex<double>, ColMajor, int> : eigamd_lu<std::complex<double> > {};
template <> struct partial_lu_impl<std::complex<float>, ColMajor, int> : eigamd_lu<std::complex<float> > {};
template <> struct partial_lu_impl<std::complex<double>, RowMajor, int> : eigamd_lu_rowmajor<std::complex<double> > {};
template <> struct partial_lu_impl<std::complex<float>, RowMajor, int> : eigamd_lu_rowmajor<std::complex<float> > {};
#endif

}  // namespace internal
}  // namespace Eigen

#endif  // EIGEN_AMD_EIGEN_SEAM_H
