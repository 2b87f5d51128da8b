/* CPU oracle for the blocked partial-pivoting LU whose trailing update is the
 * GEMM path: a plain-C restatement of partial_lu_impl::unblocked_lu / blocked_lu
 * (Eigen/src/LU/PartialPivLU.h:357-498) and of the getrf wrapper
 * (lapack/lu.cpp:14-41), for float, double and their complex counterparts (the reference's
 * functions are templates; for a complex scalar the score of an entry is its modulus, std::abs,
 * PartialPivLU.h:375), column-major.
 *
 * TEST INFRASTRUCTURE ONLY (see gemm_oracle.c).
 *
 * Restated exactly: block-size rule, two-level recursion (panel of blockSize
 * columns factorised with blocks of 16, base case at <= 16), pivot choice (first
 * row of largest |a|), zero-pivot bookkeeping, the order in which row
 * transpositions are applied to the left and right parts, transposition and
 * first_zero_pivot outputs.  Not restated instruction for instruction: the
 * triangular solve A12 = L11^-1 A12 is plain forward substitution (the reference
 * runs its blocked triangular_solve_matrix, TriangularSolverMatrix.h:40-), and
 * the trailing update calls the GEMM oracle of gemm_oracle.c; results agree with
 * the reference to rounding, pivots exactly (tests/test_lu_oracle.py).
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>

void oracle_product64_d(int, int, int, int, long, long, long, const double *, const double *, long, const double *,
                        long, const double *, double *, long);
void oracle_product64_s(int, int, int, int, long, long, long, const float *, const float *, long, const float *,
                        long, const float *, float *, long);

void oracle_product64_c(int, int, int, int, long, long, long, const float *, const float *, long, const float *,
                        long, const float *, float *, long);
void oracle_product64_z(int, int, int, int, long, long, long, const double *, const double *, long, const double *,
                        long, const double *, double *, long);

/* T: the scalar; R: the real type of a score and of the pointers the product oracle takes */
#define DEFINE_LU(T, R, SFX, ABS, PRODUCT)                                                                      \
  /* unblocked_lu, PartialPivLU.h:357-412 (dynamic-size branch: endk = size) */                                \
  static long oracle_unblocked_lu_##SFX(long rows, long cols, T *lu, long ld, int *transp, int *nb_transp) {   \
    const long size = rows < cols ? rows : cols;                                                               \
    long first_zero_pivot = -1;                                                                                \
    *nb_transp = 0;                                                                                            \
    for (long k = 0; k < size; ++k) {                                                                          \
      long piv = k;                                                                                            \
      R biggest = ABS(lu[k + k * ld]);                                                                         \
      for (long i = k + 1; i < rows; ++i)      /* maxCoeff: first index of the maximum */                      \
        if (ABS(lu[i + k * ld]) > biggest) { biggest = ABS(lu[i + k * ld]); piv = i; }                         \
      transp[k] = (int)piv;                                                                                    \
      if (biggest != (R)0) {                                                                                   \
        if (piv != k) {                                                                                        \
          for (long j = 0; j < cols; ++j) { T t = lu[k + j * ld]; lu[k + j * ld] = lu[piv + j * ld]; lu[piv + j * ld] = t; } \
          ++*nb_transp;                                                                                        \
        }                                                                                                      \
        const T d = lu[k + k * ld];                                                                            \
        for (long i = k + 1; i < rows; ++i) lu[i + k * ld] /= d;                                               \
      } else if (first_zero_pivot == -1) {                                                                     \
        first_zero_pivot = k;                                                                                  \
      }                                                                                                        \
      if (k < rows - 1)                                                                                        \
        for (long j = k + 1; j < cols; ++j) {                                                                  \
          const T u = lu[k + j * ld];                                                                          \
          for (long i = k + 1; i < rows; ++i) lu[i + j * ld] -= lu[i + k * ld] * u;                            \
        }                                                                                                      \
    }                                                                                                          \
    return first_zero_pivot;                                                                                   \
  }                                                                                                            \
  /* blocked_lu, PartialPivLU.h:429-498 */                                                                     \
  long oracle_blocked_lu_##SFX(long rows, long cols, T *lu, long ld, int *transp, int *nb_transp,              \
                               long max_block_size) {                                                          \
    const long size = rows < cols ? rows : cols;                                                               \
    if (size <= 16) return oracle_unblocked_lu_##SFX(rows, cols, lu, ld, transp, nb_transp);                   \
    long block_size = size / 8;                                                                                \
    block_size = (block_size / 16) * 16;                                                                       \
    if (block_size < 8) block_size = 8;                                                                        \
    if (block_size > max_block_size) block_size = max_block_size;                                              \
    *nb_transp = 0;                                                                                            \
    long first_zero_pivot = -1;                                                                                \
    for (long k = 0; k < size; k += block_size) {                                                              \
      const long bs = size - k < block_size ? size - k : block_size;                                           \
      const long trows = rows - k - bs, tsize = size - k - bs;                                                 \
      int nb_in_panel = 0;                                                                                     \
      const long ret = oracle_blocked_lu_##SFX(trows + bs, bs, lu + k + k * ld, ld, transp + k, &nb_in_panel, 16); \
      if (ret >= 0 && first_zero_pivot == -1) first_zero_pivot = k + ret;                                      \
      *nb_transp += nb_in_panel;                                                                               \
      for (long i = k; i < k + bs; ++i) {      /* A_0: columns [0, k) */                                       \
        const long piv = (transp[i] += (int)k);                                                                \
        for (long j = 0; j < k; ++j) { T t = lu[i + j * ld]; lu[i + j * ld] = lu[piv + j * ld]; lu[piv + j * ld] = t; } \
      }                                                                                                        \
      if (trows) {                                                                                             \
        for (long i = k; i < k + bs; ++i) {    /* A_2: columns [k+bs, k+bs+tsize) */                           \
          const long piv = transp[i];                                                                          \
          for (long j = k + bs; j < k + bs + tsize; ++j) { T t = lu[i + j * ld]; lu[i + j * ld] = lu[piv + j * ld]; lu[piv + j * ld] = t; } \
        }                                                                                                      \
        /* A12 = A11^-1 A12, A11 unit lower (:492) */                                                          \
        for (long j = k + bs; j < k + bs + tsize; ++j)                                                         \
          for (long i = 0; i < bs; ++i) {                                                                      \
            const T x = lu[(k + i) + j * ld];                                                                  \
            for (long r = i + 1; r < bs; ++r) lu[(k + r) + j * ld] -= lu[(k + r) + (k + i) * ld] * x;          \
          }                                                                                                    \
        /* A22 -= A21 * A12 (:494): the GEMM hot path, alpha = -1 */                                           \
        const T minus_one = (T)-1, one = (T)1;                                                                 \
        PRODUCT(0, 0, 0, 0, trows, tsize, bs, (const R *)&minus_one, (const R *)(lu + (k + bs) + k * ld), ld,  \
                (const R *)(lu + k + (k + bs) * ld), ld, (const R *)&one, (R *)(lu + (k + bs) + (k + bs) * ld), ld); \
      }                                                                                                        \
    }                                                                                                          \
    return first_zero_pivot;                                                                                   \
  }                                                                                                            \
  /* getrf, lapack/lu.cpp:14-41: 1-based ipiv, info = first zero pivot + 1 */                                  \
  int oracle_getrf_##SFX(const int *m, const int *n, T *a, const int *lda, int *ipiv, int *info) {             \
    *info = 0;                                                                                                 \
    if (*m < 0) *info = -1;                                                                                    \
    else if (*n < 0) *info = -2;                                                                               \
    else if (*lda < (*m > 1 ? *m : 1)) *info = -4;                                                             \
    if (*info != 0) return 0;                                                                                  \
    if (*m == 0 || *n == 0) return 0;                                                                          \
    int nb = 0;                                                                                                \
    const long ret = oracle_blocked_lu_##SFX(*m, *n, a, *lda, ipiv, &nb, 256);                                 \
    for (int i = 0; i < (*m < *n ? *m : *n); ++i) ipiv[i]++;                                                   \
    if (ret >= 0) *info = (int)ret + 1;                                                                        \
    return 0;                                                                                                  \
  }

DEFINE_LU(double, double, d, fabs, oracle_product64_d)
DEFINE_LU(float, float, s, fabsf, oracle_product64_s)
DEFINE_LU(float _Complex, float, c, cabsf, oracle_product64_c)
DEFINE_LU(double _Complex, double, z, cabs, oracle_product64_z)
