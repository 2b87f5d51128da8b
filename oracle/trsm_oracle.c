/* CPU oracle for the triangular solve with many right-hand sides (?trsm_), the symmetric
 * rank-k update (?syrk_) and the products with a symmetric / Hermitian / triangular operand
 * (?symm_, ?hemm_, ?trmm_): plain C, float, double, complex float, complex double, every form.
 *
 * TEST INFRASTRUCTURE ONLY.  Callers: tests/.  The product library never links this file.
 *
 * Parity pin: checked in tests/test_trsm_oracle.py and tests/test_syrk_oracle.py against (a)
 * the Netlib level-3 known-answer drivers ?CHK3 / ?CHK4 restated in netlib_l3.c with the parameter files of
 * blas/testing/?blat3.dat, (b) golden outputs of Eigen's own ?trsm_ / ?syrk_ committed under
 * tests/golden/, and (c) when oracle/_ref is built, the reference library itself on fresh
 * inputs.  Agreement with the reference is to rounding (the reference evaluates the same
 * recurrences in panels), not bitwise.
 */
#include <complex.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_NOTR 0
#define ORACLE_TR 1
#define ORACLE_ADJ 2
#define ORACLE_INVALID 0xff

static int oracle_op(char x) { /* OP() of blas/common.h:47-50 */
  if (x == 'N' || x == 'n') return ORACLE_NOTR;
  if (x == 'T' || x == 't') return ORACLE_TR;
  if (x == 'C' || x == 'c') return ORACLE_ADJ;
  return ORACLE_INVALID;
}
static int imax(int a, int b) { return a > b ? a : b; }

extern int xerbla_(const char *msg, int *info, int len);

#define SCALAR float
#define REAL float
#define SFX s
#define IS_CPLX 0
#define CONJ(x) (x)
#define LETTER 'S'
#include "trsm_oracle_impl.inc"
#undef SCALAR
#undef REAL
#undef SFX
#undef IS_CPLX
#undef CONJ
#undef LETTER

#define SCALAR double
#define REAL double
#define SFX d
#define IS_CPLX 0
#define CONJ(x) (x)
#define LETTER 'D'
#include "trsm_oracle_impl.inc"
#undef SCALAR
#undef REAL
#undef SFX
#undef IS_CPLX
#undef CONJ
#undef LETTER

#define SCALAR float _Complex
#define REAL float
#define SFX c
#define IS_CPLX 1
#define CONJ(x) conjf(x)
#define CREAL_(x) ((float _Complex)crealf(x))
#define LETTER 'C'
#include "trsm_oracle_impl.inc"
#undef CREAL_
#undef SCALAR
#undef REAL
#undef SFX
#undef IS_CPLX
#undef CONJ
#undef LETTER

#define SCALAR double _Complex
#define REAL double
#define SFX z
#define IS_CPLX 1
#define CONJ(x) conj(x)
#define CREAL_(x) ((double _Complex)creal(x))
#define LETTER 'Z'
#include "trsm_oracle_impl.inc"
