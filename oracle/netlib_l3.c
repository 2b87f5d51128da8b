/* Restatement in C of the GEMM and TRSM parts of the Netlib level-3 BLAS testers that the
 * reference ships as Fortran (blas/testing/{s,d,c,z}blat3.f; no Fortran compiler
 * exists in this image): the deterministic generators ?BEG, ?MAKE('GE'), the
 * result checker ?MMCH, the drivers ?CHK1 (GEMM) and ?CHK3 (TRSM rows, with ?MAKE('TR'))
 * and the GEMM and TRSM rows of the error-exit table ?CHKE.  The routine under test is
 * passed in as a function pointer with the F77 ?gemm_ / ?trsm_ signature.
 *
 * TEST INFRASTRUCTURE ONLY (see gemm_oracle.c).
 */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#define NMAX 65 /* dblat3.f:100 */

typedef int (*gemm_fn_t)(const char *, const char *, const int *, const int *, const int *, const void *,
                         const void *, const int *, const void *, const int *, const void *, void *,
                         const int *);

typedef int (*trsm_fn_t)(const char *, const char *, const char *, const char *, const int *, const int *, const void *,
                         const void *, const int *, void *, const int *);

typedef int (*syrk_fn_t)(const char *, const char *, const int *, const int *, const void *, const void *, const int *,
                         const void *, void *, const int *);

typedef int (*symm_fn_t)(const char *, const char *, const int *, const int *, const void *, const void *, const int *,
                         const void *, const int *, const void *, void *, const int *);

struct netlib_report {
  int ncalls;        /* calls made by ?CHK1 */
  int nfail_same;    /* an argument or C's padding came back changed */
  int nfail_ratio;   /* ?MMCH declared the result less than half accurate */
  int nfail_xerbla;  /* error exit taken on valid arguments */
  int fatal;
  double errmax;     /* largest test ratio seen (pass: < THRESH = 16, ?blat3.dat line 8) */
};

struct beg_state { int reset, i, j, ic, mi, mj; };

/* CHKXER / XERBLA of the testers (dblat3.f:2787-2870): the harness's xerbla_
 * records what the routine reported instead of printing. */
int netlib_xerbla_count = 0;
int netlib_xerbla_info = 0;
char netlib_xerbla_name[8] = {0};

int xerbla_(const char *msg, int *info, int len) {
  netlib_xerbla_count += 1;
  netlib_xerbla_info = *info;
  int l = len < 7 ? len : 7;
  memcpy(netlib_xerbla_name, msg, (size_t)l);
  netlib_xerbla_name[l] = 0;
  return 0;
}
int netlib_last_xerbla(char *name8) {
  memcpy(name8, netlib_xerbla_name, 8);
  return netlib_xerbla_count ? netlib_xerbla_info : 0;
}

#define SCALAR float
#define REAL float
#define SFX s
#define IS_CPLX 0
#define CONJ(x) (x)
#define ABS1(x) fabsf(x)
#define EPS 1.1920928955078125e-07
#include "netlib_l3_impl.inc"
#undef SCALAR
#undef REAL
#undef SFX
#undef IS_CPLX
#undef CONJ
#undef ABS1
#undef EPS

#define SCALAR double
#define REAL double
#define SFX d
#define IS_CPLX 0
#define CONJ(x) (x)
#define ABS1(x) fabs(x)
#define EPS 2.220446049250313e-16
#include "netlib_l3_impl.inc"
#undef SCALAR
#undef REAL
#undef SFX
#undef IS_CPLX
#undef CONJ
#undef ABS1
#undef EPS

#define SCALAR float _Complex
#define REAL float
#define SFX c
#define IS_CPLX 1
#define CONJ(x) conjf(x)
#define ABS1(x) (fabsf(crealf(x)) + fabsf(cimagf(x)))
#define EPS 1.1920928955078125e-07
#include "netlib_l3_impl.inc"
#undef SCALAR
#undef REAL
#undef SFX
#undef IS_CPLX
#undef CONJ
#undef ABS1
#undef EPS

#define SCALAR double _Complex
#define REAL double
#define SFX z
#define IS_CPLX 1
#define CONJ(x) conj(x)
#define ABS1(x) (fabs(creal(x)) + fabs(cimag(x)))
#define EPS 2.220446049250313e-16
#include "netlib_l3_impl.inc"

/* GEMM rows of ?CHKE (dblat3.f:1889-1985): each call must report exactly INFOT
 * through xerbla_ with the routine's name, and return.  Operands are 2x1 dummies
 * as in the tester.  Returns the number of rows that failed. */
int netlib_chke_gemm(gemm_fn_t gemm, char letter) {
  static const struct { int infot; char ta, tb; int m, n, k, lda, ldb, ldc; } rows[] = {
      {1, '/', 'N', 0, 0, 0, 1, 1, 1},  {1, '/', 'T', 0, 0, 0, 1, 1, 1},
      {2, 'N', '/', 0, 0, 0, 1, 1, 1},  {2, 'T', '/', 0, 0, 0, 1, 1, 1},
      {3, 'N', 'N', -1, 0, 0, 1, 1, 1}, {3, 'N', 'T', -1, 0, 0, 1, 1, 1},
      {3, 'T', 'N', -1, 0, 0, 1, 1, 1}, {3, 'T', 'T', -1, 0, 0, 1, 1, 1},
      {4, 'N', 'N', 0, -1, 0, 1, 1, 1}, {4, 'N', 'T', 0, -1, 0, 1, 1, 1},
      {4, 'T', 'N', 0, -1, 0, 1, 1, 1}, {4, 'T', 'T', 0, -1, 0, 1, 1, 1},
      {5, 'N', 'N', 0, 0, -1, 1, 1, 1}, {5, 'N', 'T', 0, 0, -1, 1, 1, 1},
      {5, 'T', 'N', 0, 0, -1, 1, 1, 1}, {5, 'T', 'T', 0, 0, -1, 1, 1, 1},
      {8, 'N', 'N', 2, 0, 0, 1, 1, 2},  {8, 'N', 'T', 2, 0, 0, 1, 1, 2},
      {8, 'T', 'N', 0, 0, 2, 1, 2, 1},  {8, 'T', 'T', 0, 0, 2, 1, 1, 1},
      {10, 'N', 'N', 0, 0, 2, 1, 1, 1}, {10, 'T', 'N', 0, 0, 2, 2, 1, 1},
      {10, 'N', 'T', 0, 2, 0, 1, 1, 1}, {10, 'T', 'T', 0, 2, 0, 1, 1, 1},
      {13, 'N', 'N', 2, 0, 0, 2, 1, 1}, {13, 'N', 'T', 2, 0, 0, 2, 1, 1},
      {13, 'T', 'N', 2, 0, 0, 1, 1, 1}, {13, 'T', 'T', 2, 0, 0, 1, 1, 1},
  };
  double a[8] = {0}, b[8] = {0}, c[8] = {0}, alpha[2] = {1, 0}, beta[2] = {2, 0};
  float alphaf[2] = {1, 0}, betaf[2] = {2, 0};
  const int single = letter == 'S' || letter == 'C';
  char want[8];
  snprintf(want, sizeof want, "%cGEMM ", letter);
  int nfail = 0;
  for (unsigned r = 0; r < sizeof rows / sizeof rows[0]; ++r) {
    netlib_xerbla_count = 0;
    netlib_xerbla_info = 0;
    gemm(&rows[r].ta, &rows[r].tb, &rows[r].m, &rows[r].n, &rows[r].k, single ? (void *)alphaf : (void *)alpha, a,
         &rows[r].lda, b, &rows[r].ldb, single ? (void *)betaf : (void *)beta, c, &rows[r].ldc);
    if (netlib_xerbla_count != 1 || netlib_xerbla_info != rows[r].infot || strncmp(netlib_xerbla_name, want, 6) != 0)
      nfail += 1;
  }
  netlib_xerbla_count = 0;
  return nfail;
}

/* TRSM rows of ?CHKE (dblat3.f:2150-2260): as above, 36 calls. */
int netlib_chke_trsm(trsm_fn_t trsm, char letter) {
  static const struct { int infot; char sd, ul, tr, dg; int m, n, lda, ldb; } rows[] = {
      {1, '/', 'U', 'N', 'N', 0, 0, 1, 1},  {2, 'L', '/', 'N', 'N', 0, 0, 1, 1},
      {3, 'L', 'U', '/', 'N', 0, 0, 1, 1},  {4, 'L', 'U', 'N', '/', 0, 0, 1, 1},
      {5, 'L', 'U', 'N', 'N', -1, 0, 1, 1}, {5, 'L', 'U', 'T', 'N', -1, 0, 1, 1},
      {5, 'R', 'U', 'N', 'N', -1, 0, 1, 1}, {5, 'R', 'U', 'T', 'N', -1, 0, 1, 1},
      {5, 'L', 'L', 'N', 'N', -1, 0, 1, 1}, {5, 'L', 'L', 'T', 'N', -1, 0, 1, 1},
      {5, 'R', 'L', 'N', 'N', -1, 0, 1, 1}, {5, 'R', 'L', 'T', 'N', -1, 0, 1, 1},
      {6, 'L', 'U', 'N', 'N', 0, -1, 1, 1}, {6, 'L', 'U', 'T', 'N', 0, -1, 1, 1},
      {6, 'R', 'U', 'N', 'N', 0, -1, 1, 1}, {6, 'R', 'U', 'T', 'N', 0, -1, 1, 1},
      {6, 'L', 'L', 'N', 'N', 0, -1, 1, 1}, {6, 'L', 'L', 'T', 'N', 0, -1, 1, 1},
      {6, 'R', 'L', 'N', 'N', 0, -1, 1, 1}, {6, 'R', 'L', 'T', 'N', 0, -1, 1, 1},
      {9, 'L', 'U', 'N', 'N', 2, 0, 1, 2},  {9, 'L', 'U', 'T', 'N', 2, 0, 1, 2},
      {9, 'R', 'U', 'N', 'N', 0, 2, 1, 1},  {9, 'R', 'U', 'T', 'N', 0, 2, 1, 1},
      {9, 'L', 'L', 'N', 'N', 2, 0, 1, 2},  {9, 'L', 'L', 'T', 'N', 2, 0, 1, 2},
      {9, 'R', 'L', 'N', 'N', 0, 2, 1, 1},  {9, 'R', 'L', 'T', 'N', 0, 2, 1, 1},
      {11, 'L', 'U', 'N', 'N', 2, 0, 2, 1}, {11, 'L', 'U', 'T', 'N', 2, 0, 2, 1},
      {11, 'R', 'U', 'N', 'N', 2, 0, 1, 1}, {11, 'R', 'U', 'T', 'N', 2, 0, 1, 1},
      {11, 'L', 'L', 'N', 'N', 2, 0, 2, 1}, {11, 'L', 'L', 'T', 'N', 2, 0, 2, 1},
      {11, 'R', 'L', 'N', 'N', 2, 0, 1, 1}, {11, 'R', 'L', 'T', 'N', 2, 0, 1, 1},
  };
  double a[8] = {0}, b[8] = {0}, alpha[2] = {1, 0};
  float alphaf[2] = {1, 0};
  const int single = letter == 'S' || letter == 'C';
  char want[8];
  snprintf(want, sizeof want, "%cTRSM ", letter);
  int nfail = 0;
  for (unsigned r = 0; r < sizeof rows / sizeof rows[0]; ++r) {
    netlib_xerbla_count = 0;
    netlib_xerbla_info = 0;
    trsm(&rows[r].sd, &rows[r].ul, &rows[r].tr, &rows[r].dg, &rows[r].m, &rows[r].n, single ? (void *)alphaf : (void *)alpha,
         a, &rows[r].lda, b, &rows[r].ldb);
    if (netlib_xerbla_count != 1 || netlib_xerbla_info != rows[r].infot || strncmp(netlib_xerbla_name, want, 6) != 0)
      nfail += 1;
  }
  netlib_xerbla_count = 0;
  return nfail;
}

/* HERK rows of ZCHKE / CCHKE (zblat3.f, error exits of ZHERK): 18 calls; trans 'T' is the
 * invalid value of row 2 and 'C' takes the place of 'T' elsewhere. */
typedef int (*herk_any_fn_t)(const char *, const char *, const int *, const int *, const void *, const void *, const int *,
                             const void *, void *, const int *);
int netlib_chke_herk(herk_any_fn_t herk, char letter) {
  static const struct { int infot; char ul, tr; int n, k, lda, ldc; } rows[] = {
      {1, '/', 'N', 0, 0, 1, 1},  {2, 'U', 'T', 0, 0, 1, 1},
      {3, 'U', 'N', -1, 0, 1, 1}, {3, 'U', 'C', -1, 0, 1, 1}, {3, 'L', 'N', -1, 0, 1, 1}, {3, 'L', 'C', -1, 0, 1, 1},
      {4, 'U', 'N', 0, -1, 1, 1}, {4, 'U', 'C', 0, -1, 1, 1}, {4, 'L', 'N', 0, -1, 1, 1}, {4, 'L', 'C', 0, -1, 1, 1},
      {7, 'U', 'N', 2, 0, 1, 2},  {7, 'U', 'C', 0, 2, 1, 1},  {7, 'L', 'N', 2, 0, 1, 2},  {7, 'L', 'C', 0, 2, 1, 1},
      {10, 'U', 'N', 2, 0, 2, 1}, {10, 'U', 'C', 2, 0, 1, 1}, {10, 'L', 'N', 2, 0, 2, 1}, {10, 'L', 'C', 2, 0, 1, 1},
  };
  double a[8] = {0}, c[8] = {0}, alpha = 1, beta = 2;
  float alphaf = 1, betaf = 2;
  const int single = letter == 'C';
  char want[8];
  snprintf(want, sizeof want, "%cHERK ", letter);
  int nfail = 0;
  for (unsigned r = 0; r < sizeof rows / sizeof rows[0]; ++r) {
    netlib_xerbla_count = 0;
    netlib_xerbla_info = 0;
    herk(&rows[r].ul, &rows[r].tr, &rows[r].n, &rows[r].k, single ? (void *)&alphaf : (void *)&alpha, a, &rows[r].lda,
         single ? (void *)&betaf : (void *)&beta, c, &rows[r].ldc);
    if (netlib_xerbla_count != 1 || netlib_xerbla_info != rows[r].infot || strncmp(netlib_xerbla_name, want, 6) != 0)
      nfail += 1;
  }
  netlib_xerbla_count = 0;
  return nfail;
}

/* SYR2K rows of ?CHKE (dblat3.f:2314-2390) and HER2K rows of ZCHKE: 22 calls; for HER2K 'T' is
 * the invalid trans of row 2 and 'C' takes the place of 'T' elsewhere, beta real. */
typedef int (*r2k_any_fn_t)(const char *, const char *, const int *, const int *, const void *, const void *, const int *,
                            const void *, const int *, const void *, void *, const int *);
int netlib_chke_r2k(r2k_any_fn_t r2k, char letter, int herm) {
  const char T = herm ? 'C' : 'T';
  const struct { int infot; char ul, tr; int n, k, lda, ldb, ldc; } rows[] = {
      {1, '/', 'N', 0, 0, 1, 1, 1},  {2, 'U', (char)(herm ? 'T' : '/'), 0, 0, 1, 1, 1},
      {3, 'U', 'N', -1, 0, 1, 1, 1}, {3, 'U', T, -1, 0, 1, 1, 1}, {3, 'L', 'N', -1, 0, 1, 1, 1}, {3, 'L', T, -1, 0, 1, 1, 1},
      {4, 'U', 'N', 0, -1, 1, 1, 1}, {4, 'U', T, 0, -1, 1, 1, 1}, {4, 'L', 'N', 0, -1, 1, 1, 1}, {4, 'L', T, 0, -1, 1, 1, 1},
      {7, 'U', 'N', 2, 0, 1, 1, 2},  {7, 'U', T, 0, 2, 1, 1, 1},  {7, 'L', 'N', 2, 0, 1, 1, 2},  {7, 'L', T, 0, 2, 1, 1, 1},
      {9, 'U', 'N', 2, 0, 2, 1, 2},  {9, 'U', T, 0, 2, 2, 1, 1},  {9, 'L', 'N', 2, 0, 2, 1, 2},  {9, 'L', T, 0, 2, 2, 1, 1},
      {12, 'U', 'N', 2, 0, 2, 2, 1}, {12, 'U', T, 2, 0, 1, 1, 1}, {12, 'L', 'N', 2, 0, 2, 2, 1}, {12, 'L', T, 2, 0, 1, 1, 1},
  };
  double a[8] = {0}, b[8] = {0}, c[8] = {0}, alpha[2] = {1, 0}, beta[2] = {2, 0};
  float alphaf[2] = {1, 0}, betaf[2] = {2, 0};
  const int single = letter == 'S' || letter == 'C';
  char want[8];
  snprintf(want, sizeof want, "%c%s", letter, herm ? "HER2K" : "SYR2K");
  int nfail = 0;
  for (unsigned r = 0; r < sizeof rows / sizeof rows[0]; ++r) {
    netlib_xerbla_count = 0;
    netlib_xerbla_info = 0;
    r2k(&rows[r].ul, &rows[r].tr, &rows[r].n, &rows[r].k, single ? (void *)alphaf : (void *)alpha, a, &rows[r].lda, b,
        &rows[r].ldb, single ? (void *)betaf : (void *)beta, c, &rows[r].ldc);
    if (netlib_xerbla_count != 1 || netlib_xerbla_info != rows[r].infot || strncmp(netlib_xerbla_name, want, 6) != 0)
      nfail += 1;
  }
  netlib_xerbla_count = 0;
  return nfail;
}

/* SYRK rows of ?CHKE (dblat3.f:2262-2300): 18 calls. */
int netlib_chke_syrk(syrk_fn_t syrk, char letter) {
  static const struct { int infot; char ul, tr; int n, k, lda, ldc; } rows[] = {
      {1, '/', 'N', 0, 0, 1, 1},  {2, 'U', '/', 0, 0, 1, 1},
      {3, 'U', 'N', -1, 0, 1, 1}, {3, 'U', 'T', -1, 0, 1, 1}, {3, 'L', 'N', -1, 0, 1, 1}, {3, 'L', 'T', -1, 0, 1, 1},
      {4, 'U', 'N', 0, -1, 1, 1}, {4, 'U', 'T', 0, -1, 1, 1}, {4, 'L', 'N', 0, -1, 1, 1}, {4, 'L', 'T', 0, -1, 1, 1},
      {7, 'U', 'N', 2, 0, 1, 2},  {7, 'U', 'T', 0, 2, 1, 1},  {7, 'L', 'N', 2, 0, 1, 2},  {7, 'L', 'T', 0, 2, 1, 1},
      {10, 'U', 'N', 2, 0, 2, 1}, {10, 'U', 'T', 2, 0, 1, 1}, {10, 'L', 'N', 2, 0, 2, 1}, {10, 'L', 'T', 2, 0, 1, 1},
  };
  double a[8] = {0}, c[8] = {0}, alpha[2] = {1, 0}, beta[2] = {2, 0};
  float alphaf[2] = {1, 0}, betaf[2] = {2, 0};
  const int single = letter == 'S' || letter == 'C';
  char want[8];
  snprintf(want, sizeof want, "%cSYRK ", letter);
  int nfail = 0;
  for (unsigned r = 0; r < sizeof rows / sizeof rows[0]; ++r) {
    netlib_xerbla_count = 0;
    netlib_xerbla_info = 0;
    syrk(&rows[r].ul, &rows[r].tr, &rows[r].n, &rows[r].k, single ? (void *)alphaf : (void *)alpha, a, &rows[r].lda,
         single ? (void *)betaf : (void *)beta, c, &rows[r].ldc);
    if (netlib_xerbla_count != 1 || netlib_xerbla_info != rows[r].infot || strncmp(netlib_xerbla_name, want, 6) != 0)
      nfail += 1;
  }
  netlib_xerbla_count = 0;
  return nfail;
}
