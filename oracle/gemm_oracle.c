/* CPU oracle for the dense matrix-matrix product path: a plain-C restatement of
 * Eigen's blocked GEMM (pack_lhs / pack_rhs / gebp / 3-loop driver / BLAS
 * wrapper) for float, double, complex float and complex double.
 *
 * TEST INFRASTRUCTURE ONLY.  Callers: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py.  The product library never links this file.
 *
 * Parity pin: checked in tests/test_oracle.py against (a) the Netlib level-3
 * known-answer driver restated in netlib_l3.c with the parameter files of
 * blas/testing/?blat3.dat, (b) golden outputs of Eigen's own ?gemm_ and
 * operator* committed under tests/golden/, and (c) when oracle/_ref is built,
 * the reference library itself on fresh random inputs.
 *
 * Floating point: same algorithm, blocking and per-tile k-order as the
 * reference's non-vectorised configuration, multiply and add rounded separately
 * (build with -ffp-contract=off).  The reference's SIMD build uses other
 * register-tile sizes and FMA, so agreement is to rounding, not bitwise.
 */
#include <complex.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_NOTR 0
#define ORACLE_TR 1
#define ORACLE_ADJ 2
#define ORACLE_INVALID 0xff

/* OP() of blas/common.h:47-50 */
static int oracle_op(char x) {
  if (x == 'N' || x == 'n') return ORACLE_NOTR;
  if (x == 'T' || x == 't') return ORACLE_TR;
  if (x == 'C' || x == 'c') return ORACLE_ADJ;
  return ORACLE_INVALID;
}
static int imax(int a, int b) { return a > b ? a : b; }

/* Resolved at load time: the test harness installs a recording xerbla_
 * (netlib_l3.c); the signature is that of blas/xerbla.cpp:15. */
extern int xerbla_(const char *msg, int *info, int len);

#define SCALAR float
#define REAL float
#define SFX s
#define IS_CPLX 0
#define CONJ(x) (x)
#define LETTER 'S'
#include "gemm_oracle_impl.inc"
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
#include "gemm_oracle_impl.inc"
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
#define LETTER 'C'
#include "gemm_oracle_impl.inc"
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
#define LETTER 'Z'
#include "gemm_oracle_impl.inc"
