// Triangular solve with many right-hand sides on the device:
//   op(A) * X = alpha * B   (side = left)    or    X * op(A) = alpha * B   (side = right),
// A triangular (upper / lower, unit / non-unit diagonal), X overwrites B.
//
// Replaces triangular_solve_matrix (Eigen/src/Core/products/TriangularSolverMatrix.h:40-…)
// behind EIGEN_BLAS_FUNC(trsm) (blas/level3_impl.h:78-178).  The reference walks small
// panels with scalar substitution and updates the rest with its GEMM kernel
// (TriangularSolverMatrix.h:166,183); here the diagonal blocks (128 x 128 or 256 x 256, taken in pairs) are
// inverted by a small kernel, a block solve is ONE in-place product with that inverse, and the
// blocks are combined by recursive halving, whose updates are MFMA GEMMs with long reductions.
// As in the reference the solve comes first and alpha is applied afterwards (level3_impl.h:163-175).
//
// Multiplying by an explicit inverse is as accurate as substitution only while the inverse is
// not much larger than the triangle's own entries: its error grows with max|inv(T)| * max|T|,
// the reference's substitution stays at a few eps whatever the triangle (it is backward stable
// row by row).  So every diagonal block is measured after its inversion (trtri_decide): where
// that product exceeds kGrowthLimit (or is not finite) the block is solved by substitution on
// the device (trsm_block_subst, the reference's recurrence in 32-row steps) and its "inverse"
// is replaced by the identity, so that the launch sequence does not depend on the data and
// nothing has to come back to the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>

#include "gemm_launch.hpp"
#include "trsm.hpp"

namespace eigamd {

namespace {

constexpr int kSmallNB = 128;  // diagonal block; kLargeNB where the GEMM tile allows it (real kinds)
constexpr int kLargeNB = 256;
constexpr int kInvCols = 32;   // columns of the inverse one workgroup owns

// The in-place products below rely on one GEMM tile covering the whole block extent: along m
// (tile extent BM) for a left-side solve, along n (BN) for a right-side one.  A 256-wide block
// halves the number of leaf products and makes them whole tiles of the f64 / f32 kernels
// (m = 128 leaves ran in the bounds-checked instantiation, a fifth of a dtrsm's time).
static_assert(Geometry<F32>::BM >= kSmallNB && Geometry<F32>::BN >= kSmallNB, "tile smaller than the trsm block");
static_assert(Geometry<F64>::BM >= kSmallNB && Geometry<F64>::BN >= kSmallNB, "tile smaller than the trsm block");
static_assert(Geometry<C32>::BM >= kSmallNB && Geometry<C32>::BN >= kSmallNB, "tile smaller than the trsm block");
static_assert(Geometry<C64>::BM >= kSmallNB && Geometry<C64>::BN >= kSmallNB, "tile smaller than the trsm block");
inline int block_extent(int kind, bool left) {
  if (kind == F64) return (left ? Geometry<F64>::BM : Geometry<F64>::BN) >= kLargeNB ? kLargeNB : kSmallNB;
  if (kind == F32) return (left ? Geometry<F32>::BM : Geometry<F32>::BN) >= kLargeNB ? kLargeNB : kSmallNB;
  return kSmallNB;
}

template <typename R> struct Cplx {
  R x, y;
};
template <typename R> __device__ __forceinline__ Cplx<R> operator+(Cplx<R> a, Cplx<R> b) { return {a.x + b.x, a.y + b.y}; }
template <typename R> __device__ __forceinline__ Cplx<R> operator*(Cplx<R> a, Cplx<R> b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename R> __device__ __forceinline__ Cplx<R> operator-(Cplx<R> a) { return {-a.x, -a.y}; }
template <typename R> __device__ __forceinline__ Cplx<R> operator-(Cplx<R> a, Cplx<R> b) { return {a.x - b.x, a.y - b.y}; }
template <typename R> __device__ __forceinline__ Cplx<R> conj_of(Cplx<R> a) { return {a.x, -a.y}; }
__device__ __forceinline__ float conj_of(float a) { return a; }
__device__ __forceinline__ double conj_of(double a) { return a; }
template <typename R> __device__ __forceinline__ double mag(Cplx<R> a) { return fabs((double)a.x) + fabs((double)a.y); }
__device__ __forceinline__ double mag(float a) { return fabs((double)a); }
__device__ __forceinline__ double mag(double a) { return fabs(a); }
// Smith's division, as std::complex does it.
template <typename R> __device__ __forceinline__ Cplx<R> operator/(Cplx<R> a, Cplx<R> b) {
  if (fabs(b.x) >= fabs(b.y)) {
    const R r = b.y / b.x, d = b.x + b.y * r;
    return {(a.x + a.y * r) / d, (a.y - a.x * r) / d};
  }
  const R r = b.x / b.y, d = b.x * r + b.y;
  return {(a.x * r + a.y) / d, (a.y * r - a.x) / d};
}
template <typename T> __device__ __forceinline__ T zero_of() { return T{}; }
template <typename T> struct One;
template <> struct One<float> { static __device__ float v() { return 1.f; } };
template <> struct One<double> { static __device__ double v() { return 1.0; } };
template <> struct One<Cplx<float>> { static __device__ Cplx<float> v() { return {1.f, 0.f}; } };
template <> struct One<Cplx<double>> { static __device__ Cplx<double> v() { return {1.0, 0.0}; } };

// Inverse of every diagonal block of the na x na triangular matrix A at once.  Block b
// (rows/columns [b*kNB, b*kNB + nbb)) is read through its stored triangle only -- the other
// triangle, and the diagonal when `unit`, are never touched (the Netlib testers fill them
// with rogue values) -- and its inverse is written as a full kNB x kNB matrix (zeros in the
// other triangle) at X + b * kNB * kNB.  An upper triangle is inverted as the transpose of
// a lower one.  grid = (kNB / kInvCols, number of blocks); a workgroup owns 32 columns of
// one inverse in LDS and walks down the rows 32 at a time: the part of the inner products
// that needs finished rows, times the inverse of the 32 x 32 triangle on the diagonal (those
// inverses are formed first, all at once).
// It also notes, per block, max|inv(T_b)| and max|T_b| over what it wrote and read (gmax[2 b], gmax[2 b + 1]: the bit
// patterns of non-negative doubles order like the numbers, NaN above infinity, so an integer atomicMax folds them; the
// words are zero on entry), which is all trtri_decide needs.
__device__ __forceinline__ void note_max(unsigned long long* word, double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(fabs(v));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_down(b, off, 64);
    b = o > b ? o : b;
  }
  if ((threadIdx.x & 63) == 0 && b != 0) atomicMax(word, b);
}
// Where the image of block b lies.  Alone: X + b NB^2, leading dimension NB.  In pairs: blocks 2p and 2p + 1 are the
// diagonal quadrants of a 2 NB x 2 NB image at X + p (2 NB)^2 (leading dimension 2 NB) whose other two quadrants are
// zero and, once trtri_pair_product has run, the off-diagonal block of the inverse of the pair's 2 NB x 2 NB triangle.
template <typename T, int NB> __device__ __forceinline__ T* image_of(T* X, int b, int pairs) {
  return pairs ? X + (int64_t)(b >> 1) * (4 * NB * NB) + (b & 1) * (NB + (int64_t)NB * 2 * NB) : X + (int64_t)b * NB * NB;
}
template <typename T, int NB>
__global__ void __launch_bounds__(256) trtri_blocks(const T* __restrict__ A, int64_t lda, int na, int upper, int unit,
                                                    T* __restrict__ X, unsigned long long* __restrict__ gmax, int pairs) {
  constexpr int kNB = NB;   // block extent of this instantiation
  const int ldx = pairs ? 2 * NB : NB;
  constexpr int kSteps = NB / kInvCols;
  __shared__ T s_x[kInvCols][kNB + 1];
  __shared__ T s_acc[kInvCols][kInvCols + 1];
  __shared__ T s_dinv[kSteps][kInvCols][kInvCols + 1];  // [j][k][i]: first L(32 j + i, 32 j + k) (i >= k), then its inverse
  const int b0 = blockIdx.y * kNB;
  const int bs = min(kNB, na - b0);
  const T* Ab = A + b0 + (int64_t)b0 * lda;
  T* Xb = image_of<T, NB>(X, (int)blockIdx.y, pairs);
  auto at = [&](int i, int k) { return upper ? Ab[k + (int64_t)i * lda] : Ab[i + (int64_t)k * lda]; };  // i > k
  const int first = blockIdx.x * kInvCols;
  for (int idx = threadIdx.x; idx < kInvCols * kNB; idx += 256) s_x[idx / kNB][idx % kNB] = zero_of<T>();
  // The 32 x 32 triangles on the diagonal, from this workgroup's columns on, all at once: loaded in one batch, then each
  // inverted by half a wavefront (a column per lane, 32 steps of substitution) -- once, in parallel, instead of once per
  // 32-row step with one half-wavefront at work.  Rows past the end of a ragged block count as rows of the identity.
  for (int idx = threadIdx.x; idx < kSteps * kInvCols * kInvCols; idx += 256) {
    const int j = idx / (kInvCols * kInvCols), kk = (idx / kInvCols) % kInvCols, ii = idx % kInvCols;
    if (j < (int)blockIdx.x) continue;
    const int r = j * kInvCols + ii, c = j * kInvCols + kk;
    T v = zero_of<T>();
    if (ii == kk) v = (unit || r >= bs) ? One<T>::v() : Ab[r + (int64_t)r * lda];
    else if (ii > kk && r < bs) v = at(r, c);
    s_dinv[j][kk][ii] = v;
  }
  __syncthreads();
  {
    const int j = threadIdx.x / kInvCols, c = threadIdx.x % kInvCols;
    static_assert(kSteps <= 256 / kInvCols, "a half-wavefront per 32 x 32 triangle");
    T x[kInvCols];
    const bool mine = j < kSteps && j >= (int)blockIdx.x;
    if (mine) {
#pragma unroll
      for (int ii = 0; ii < kInvCols; ++ii) {
        T v = zero_of<T>();
#pragma unroll
        for (int kk = 0; kk < kInvCols; ++kk)
          if (kk < ii) v = v + s_dinv[j][kk][ii] * x[kk];
        if (ii < c) x[ii] = zero_of<T>();
        else if (ii == c) x[ii] = One<T>::v() / s_dinv[j][ii][ii];
        else x[ii] = -(v / s_dinv[j][ii][ii]);
      }
    }
    __syncthreads();   // every lane has read its triangle
    if (mine) {
#pragma unroll
      for (int ii = 0; ii < kInvCols; ++ii) s_dinv[j][c][ii] = x[ii];
    }
  }
  __syncthreads();
  const int il = threadIdx.x % kInvCols, cg = threadIdx.x / kInvCols;
  for (int i0 = first; i0 < bs; i0 += kInvCols) {
    const int i = i0 + il, j = i0 / kInvCols;
    T acc[4] = {zero_of<T>(), zero_of<T>(), zero_of<T>(), zero_of<T>()};
    if (i < bs) {
      for (int k = first; k < i0; k += 8) {
        T l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) l[u] = at(i, k + u);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = acc[q] + l[u] * s_x[cg * 4 + q][k + u];
      }
    }
    if (i0 == first) {   // the column block's own triangle: its inverse as it stands
#pragma unroll
      for (int q = 0; q < 4; ++q) s_x[cg * 4 + q][i0 + il] = s_dinv[j][cg * 4 + q][il];
    } else {             // rows below it: X(rows, c) = -inv(D) * (L(rows, above) X(above, c)), 32 x 32 x 32 by all threads
#pragma unroll
      for (int q = 0; q < 4; ++q) s_acc[cg * 4 + q][il] = acc[q];
      __syncthreads();
      T v[4] = {zero_of<T>(), zero_of<T>(), zero_of<T>(), zero_of<T>()};
#pragma unroll
      for (int kk = 0; kk < kInvCols; ++kk) {
        const T d = s_dinv[j][kk][il];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] + d * s_acc[cg * 4 + q][kk];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) s_x[cg * 4 + q][i0 + il] = -v[q];
    }
    __syncthreads();
  }
  // full kNB x kNB image, zeros outside the bs x bs triangle
  double mx = 0.0, ma = unit ? 1.0 : 0.0;
  for (int idx = threadIdx.x; idx < kInvCols * kNB; idx += 256) {
    const int c = idx / kNB, i = idx % kNB;
    const int g = first + c;
    const T v = (g < bs && i < bs) ? s_x[c][i] : zero_of<T>();
    if (upper) Xb[g + (int64_t)i * ldx] = v;
    else Xb[i + (int64_t)g * ldx] = v;
    const double vm = mag(v);
    mx = (vm > mx || vm != vm) ? vm : mx;
    if (g < bs && i < bs && (i > g || (i == g && !unit))) {   // the stored triangle, this workgroup's columns
      const double a = mag(i == g ? Ab[g + (int64_t)g * lda] : at(i, g));
      ma = (a > ma || a != a) ? a : ma;
    }
  }
  note_max(gmax + 2 * blockIdx.y, mx);
  note_max(gmax + 2 * blockIdx.y + 1, ma);
  // In pairs: the quadrant of the pair's image that stays zero is written by the block that shares its columns (a lower
  // triangle's inverse: rows of the first block, columns of the second) or its rows (upper).
  if (pairs && (int)(blockIdx.y & 1) == (upper ? 0 : 1)) {
    for (int idx = threadIdx.x; idx < kInvCols * kNB; idx += 256) {
      const int c = idx / kNB, i = idx % kNB;
      const int g = first + c;
      if (upper) Xb[(kNB + i) + (int64_t)g * ldx] = zero_of<T>();   // (second block's rows, this block's columns)
      else Xb[(i - kNB) + (int64_t)g * ldx] = zero_of<T>();         // (first block's rows, this block's columns)
    }
  }
}

// One stage of the off-diagonal block of a pair's inverse, for all pairs in one launch: C_p = (-)P_p Q_p, NB x NB x NB
// each, operands `sP`, `sQ`, `sC` elements apart from pair to pair.  inv([T11 0; T21 T22]) has -inv(T22) T21 inv(T11)
// below the diagonal: stage 1 forms T21 inv(T11) into scratch, stage 2 multiplies by -inv(T22) into the image (an upper
// triangle likewise above the diagonal).  The images are whole NB x NB blocks with zeros where the triangle ends, so
// only the block of A needs its extent: rows >= p_rows or columns >= p_cols of P (and likewise of Q) read as zero in the
// LAST pair, which is the one a ragged matrix cuts short.  grid = ((NB / 32)^2, pairs); a workgroup owns a 32 x 32 tile
// of C (thread: rows ti, ti + 16, columns 2 tj, 2 tj + 1) and walks k 32 at a time, one chunk of loads ahead.
// p_tri / q_tri: that operand is a triangle (1 lower, 2 upper; an inverse's image), and the part of the k range where it
// is zero for the whole tile is not walked.
template <typename T>
__global__ void __launch_bounds__(256) trtri_pair_product(const T* __restrict__ P, int64_t ldp, int64_t sP, int p_rows, int p_cols,
                                                          const T* __restrict__ Q, int64_t ldq, int64_t sQ, int q_rows, int q_cols,
                                                          T* __restrict__ C, int64_t ldc, int64_t sC, int nb, int negate,
                                                          int p_tri, int q_tri, const int* __restrict__ pair_flags) {
  constexpr int kT = 32, kK = 32, kPer = kT * kK / 256;
  __shared__ T s_a[kK][kT + 1];   // s_a[kk][i] = P(r0 + i, k0 + kk)
  __shared__ T s_b[kK][kT + 2];   // s_b[kk][j] = Q(k0 + kk, c0 + j)
  const int pair = blockIdx.y;
  const bool last = pair == (int)gridDim.y - 1;
  const int pr = last ? p_rows : nb, pc = last ? p_cols : nb, qr = last ? q_rows : nb, qc = last ? q_cols : nb;
  const int per_row = nb / kT;
  const int r0 = ((int)blockIdx.x % per_row) * kT, c0 = ((int)blockIdx.x / per_row) * kT;
  P += (int64_t)pair * sP;
  Q += (int64_t)pair * sQ;
  C += (int64_t)pair * sC;
  const int t = threadIdx.x, ti = t % 16, tj = t / 16;
  // loads of a chunk: P 32 x kK and Q kK x 32, kPer each per thread, consecutive threads down a column
  constexpr int kAStep = 256 / kT, kBStep = 256 / kK;
  const int ai = t % kT, ak = t / kT;          // and ak + kAStep u
  const int bk = t % kK, bj = t / kK;          // and bj + kBStep u
  T ra[kPer], rb[kPer];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int row = r0 + ai, col = k0 + ak + kAStep * u;
      ra[u] = (row < pr && col < pc) ? P[row + (int64_t)col * ldp] : zero_of<T>();
      const int qrow = k0 + bk, qcol = c0 + bj + kBStep * u;
      rb[u] = (qrow < qr && qcol < qc) ? Q[qrow + (int64_t)qcol * ldq] : zero_of<T>();
    }
  };
  T acc[2][2] = {{zero_of<T>(), zero_of<T>()}, {zero_of<T>(), zero_of<T>()}};
  int k_begin = 0, k_end = nb;
  if (p_tri == 1) k_end = min(k_end, r0 + kT);      // P(i, k) = 0 for k > i
  if (p_tri == 2) k_begin = max(k_begin, r0);       // ... for k < i
  if (q_tri == 1) k_begin = max(k_begin, c0);       // Q(k, j) = 0 for k < j
  if (q_tri == 2) k_end = min(k_end, c0 + kT);      // ... for k > j
  fetch(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += kK) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      s_a[ak + kAStep * u][ai] = ra[u];
      s_b[bk][bj + kBStep * u] = rb[u];
    }
    __syncthreads();
    if (k0 + kK < k_end) fetch(k0 + kK);
#pragma unroll
    for (int kk = 0; kk < kK; ++kk) {
      const T a0 = s_a[kk][ti], a1 = s_a[kk][ti + 16];
      const T b0 = s_b[kk][2 * tj], b1 = s_b[kk][2 * tj + 1];
      acc[0][0] = acc[0][0] + a0 * b0; acc[0][1] = acc[0][1] + a0 * b1;
      acc[1][0] = acc[1][0] + a1 * b0; acc[1][1] = acc[1][1] + a1 * b1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      T v = negate ? -acc[q][e] : acc[q][e];
      if (pair_flags && pair_flags[2 * pair]) v = zero_of<T>();   // a pair that is solved by substitution
      C[(r0 + ti + 16 * q) + (int64_t)(c0 + 2 * tj + e) * ldc] = v;
    }
}


// After trtri_blocks: flags[b] = 1 when max|inv(T_b)| * max|T_b| exceeds `limit` or is not finite;
// the inverse of such a block is replaced by the identity (the block is solved by substitution).
// grid = number of blocks, 256 threads.
template <typename T, int NB>
__global__ void __launch_bounds__(256) trtri_decide(T* __restrict__ X, double limit, const unsigned long long* __restrict__ gmax,
                                                    int* __restrict__ flags, int pairs) {
  T* Xb = image_of<T, NB>(X, (int)blockIdx.x, pairs);
  const int ldx = pairs ? 2 * NB : NB;
  __shared__ int s_subst;
  if (threadIdx.x == 0) {
    auto grown = [&](int b) {
      const double mx = __longlong_as_double((long long)gmax[2 * b]);
      const double ma = __longlong_as_double((long long)gmax[2 * b + 1]);   // (at least 1 for a unit diagonal)
      return !(mx * ma <= limit);
    };
    // In pairs the two blocks of a pair are decided together: one of them over the limit and the whole pair is solved
    // by substitution (both images the identity, the off-diagonal block zero: trtri_pair_product looks at the flag).
    const int partner = (int)blockIdx.x ^ 1;
    s_subst = grown((int)blockIdx.x) || (pairs && partner < (int)gridDim.x && grown(partner));
    flags[blockIdx.x] = s_subst;
  }
  __syncthreads();
  const bool subst = s_subst != 0;
  if (subst)
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256)
      Xb[idx % NB + (int64_t)(idx / NB) * ldx] = (idx % NB == idx / NB) ? One<T>::v() : zero_of<T>();
}

// One diagonal block by substitution, in place, for the blocks trtri_decide flagged (others: the
// kernel returns at once).  Canonical form: M x = b with M lower triangular of extent pb, one thread
// per right-hand side; the unknowns are taken 32 at a time (the 32 x 32 triangle from LDS, then
// the rows below get the chunk's contribution with one read-modify-write each).  The forms of
// ?trsm_ map onto it through the index arithmetic of m_at():
//   right side  X op(A) = B  is  op(A)^T X^T = B^T  (a right-hand side is a row of B),
//   an upper triangular M is a lower one with its indices reversed (`rev`).
// inc_i: distance in B between consecutive unknowns of one right-hand side, inc_c: between
// right-hand sides.  pb may be two blocks (a pair): the columns of M that a chunk needs pass through LDS in windows of
// kLargeNB rows.
constexpr int kSubChunk = 32;
template <typename T>
__global__ void __launch_bounds__(256) trsm_block_subst(const int* __restrict__ flag, const T* __restrict__ A, int64_t lda, int pb,
                                                        int rev, int right, int trans, int conj, int unit, T* __restrict__ B,
                                                        int64_t inc_i, int64_t inc_c, int nrhs) {
  if (*flag == 0) return;
  __shared__ T s_t[kSubChunk][kLargeNB + 1];  // s_t[kk][i - i0] = M(i, i0 + kk)
  auto m_at = [&](int i, int k) -> T {        // canonical indices, i >= k
    if (i == k && unit) return One<T>::v();
    int r = rev ? pb - 1 - i : i, c = rev ? pb - 1 - k : k;
    if (right) { const int t = r; r = c; c = t; }
    if (trans) { const int t = r; r = c; c = t; }
    const T v = A[r + (int64_t)c * lda];
    return conj ? conj_of(v) : v;
  };
  const int col = blockIdx.x * 256 + threadIdx.x;
  const bool active = col < nrhs;
  T* b = B + (int64_t)col * inc_c;
  auto b_at = [&](int i) -> T& { return b[(int64_t)(rev ? pb - 1 - i : i) * inc_i]; };
  for (int i0 = 0; i0 < pb; i0 += kSubChunk) {
    const int w = min(kSubChunk, pb - i0);
    T x[kSubChunk];
    for (int base = 0; base < pb - i0; base += kLargeNB) {   // rows [i0 + base, i0 + top) of the chunk's columns
      const int top = min(base + kLargeNB, pb - i0);
      for (int idx = threadIdx.x; idx < kSubChunk * (top - base); idx += 256) {
        const int kk = idx % kSubChunk, ii = base + idx / kSubChunk;
        s_t[kk][ii - base] = (kk < w && ii >= kk) ? m_at(i0 + ii, i0 + kk) : (kk == ii ? One<T>::v() : zero_of<T>());
      }
      __syncthreads();
      if (active) {
        if (base == 0) {
#pragma unroll
          for (int ii = 0; ii < kSubChunk; ++ii) x[ii] = ii < w ? b_at(i0 + ii) : zero_of<T>();
#pragma unroll
          for (int ii = 0; ii < kSubChunk; ++ii) {
            T v = x[ii];
#pragma unroll
            for (int kk = 0; kk < kSubChunk; ++kk)
              if (kk < ii) v = v - s_t[kk][ii] * x[kk];
            x[ii] = ii < w ? v / s_t[ii][ii] : zero_of<T>();
          }
#pragma unroll
          for (int ii = 0; ii < kSubChunk; ++ii)
            if (ii < w) b_at(i0 + ii) = x[ii];
        }
        for (int i = i0 + max(base, kSubChunk); i < i0 + top; ++i) {
          T acc = zero_of<T>();
#pragma unroll
          for (int kk = 0; kk < kSubChunk; ++kk) acc = acc + s_t[kk][i - i0 - base] * x[kk];
          b_at(i) = b_at(i) - acc;
        }
      }
      __syncthreads();
    }
  }
}

// max|inv(T)| * max|T| above which a diagonal block is solved by substitution.  The error of the
// product with the inverse is about that many times a substitution's; 8 keeps it inside what the
// parity tests allow for rounding (test ratio 16).  EIGAMD_TRSM_GROWTH_LIMIT overrides it (0: every
// block by substitution; inf: none), read at every call.
constexpr double kGrowthLimit = 8.0;
inline double growth_limit() {
  if (const char* e = std::getenv("EIGAMD_TRSM_GROWTH_LIMIT")) return std::atof(e);
  return kGrowthLimit;
}

hipError_t gemm(int kind, int opa, int opb, int64_t m, int64_t n, int64_t k, double alpha, const void* A, int64_t lda,
                const void* B, int64_t ldb, double beta, void* C, int64_t ldc, hipStream_t s, void* partials, size_t partials_bytes,
                void* split_dest = nullptr, int64_t split_ldd = 0, bool* split_dest_used = nullptr) {
  if (m <= 0 || n <= 0) return hipSuccess;
  GemmCall c;
  c.kind = kind;
  c.opa = opa; c.opb = opb;
  c.m = m; c.n = n; c.k = k;
  c.alpha[0] = alpha; c.alpha[1] = 0.0;
  c.beta[0] = beta; c.beta[1] = 0.0;
  c.A = A; c.lda = lda;
  c.B = B; c.ldb = ldb;
  c.C = C; c.ldc = ldc;
  c.split_dest = split_dest; c.split_ldd = split_ldd; c.split_dest_used = split_dest_used;
  c.partials = partials; c.partials_bytes = partials_bytes; c.partials_keep_helpers = true;
  switch (kind) {
    case F32: return launch_gemm_f32(c, s);
    case F64: return launch_gemm_f64(c, s);
    case C32: return launch_gemm_c32(c, s);
    default:  return launch_gemm_c64(c, s);
  }
}

constexpr int kElem[4] = {4, 8, 8, 16};

struct InvScratch {
  std::mutex mu;
  void* buf = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;  // last use of buf
};
InvScratch& inv_scratch() {
  static InvScratch s[kMaxDevices];
  return s[current_device_slot()];
}

#define TR_TRY(expr)                 \
  do {                               \
    hipError_t e_ = (expr);          \
    if (e_ != hipSuccess) return e_; \
  } while (0)

}  // namespace

hipError_t trsm_device(int kind, bool left, bool upper, int op, bool unit, int64_t m, int64_t n, const double alpha[2],
                       const void* A, int64_t lda, void* B, int64_t ldb, hipStream_t s) {
  if (m <= 0 || n <= 0) return hipSuccess;
  const int es = kElem[kind];
  const int64_t na = left ? m : n;
  // A leaf B_j <- inv_j * B_j (or B_j * inv_j) may run in place where one GEMM tile covers the block's extent (a
  // workgroup then overwrites only what it alone has read), which ties the block to the large tile: 128 columns on
  // the right side of a double solve, and a launch of few large tiles for every leaf.  Instead the leaf's result
  // goes to a scratch image and is copied back: any tile may run the leaf (the launcher takes the small one for
  // these few-tile products) and the blocks are 256 wide on either side for the real kinds.  Measured, in place ->
  // through the image: dtrsm right side 4096 23.7 -> 32.5 TFLOP/s, 8192 47.3 -> 57.1, 16384 62.8 -> 66.0; left side
  // 28.9 -> 30.5, 54.9 -> 56.0, 64.8 -> 65.5; strsm 4096 32 -> 40, 8192 78 -> 90; ztrsm 2048 20 -> 24, 4096 44 -> 49.
  // EIGAMD_TRSM_LEAF_COPY=0: in place, with the block extents that needs (measurements).
  static const bool leaf_copy = [] { const char* e = getenv("EIGAMD_TRSM_LEAF_COPY"); return !(e && atoi(e) == 0); }();
  const int64_t kNB = (leaf_copy && (kind == F64 || kind == F32)) ? kLargeNB : block_extent(kind, left);
  const int64_t nblk = (na + kNB - 1) / kNB;
  // Leaves of TWO diagonal blocks: the inverse of a pair's 2 kNB x 2 kNB triangle is put together from the two
  // blocks' inverses (-inv(T22) T21 inv(T11) off the diagonal, two small products for all pairs at once:
  // trtri_pair_product), and a leaf is one product with it where there were two leaves and the update between them --
  // three products of 256 x nrhs x 256, each a launch of few tiles with a cut reduction, become one of 512 x nrhs x 512.
  // The growth of the two blocks' inverses is judged per pair: one of them over the limit and the pair's image is the
  // identity and the whole pair is solved by substitution in front of the product.  EIGAMD_TRSM_PAIRS=0: leaves of one
  // block.
  const char* pairs_env = getenv("EIGAMD_TRSM_PAIRS");   // (read at every call: tests compare the two forms)
  const bool pairs_wanted = !(pairs_env && atoi(pairs_env) == 0);
  const bool pairs = leaf_copy && pairs_wanted && nblk >= 2;
  const int64_t unit_blocks = pairs ? 2 : 1, kUnit = unit_blocks * kNB;
  const int64_t nunits = (nblk + unit_blocks - 1) / unit_blocks;
  const char* Ab = static_cast<const char*>(A);
  char* Bb = static_cast<char*>(B);
  auto a_at = [&](int64_t r, int64_t c) { return Ab + (r + c * lda) * es; };
  auto b_at = [&](int64_t r, int64_t c) { return Bb + (r + c * ldb) * es; };

  // inverses of all diagonal blocks (scratch shared by all callers: a use waits for the last one)
  InvScratch& sc = inv_scratch();
  std::lock_guard<std::mutex> lock(sc.mu);
  const size_t pair_tmp_bytes = pairs ? (size_t)nunits * kNB * kNB * es : 0;   // T21 inv(T11) of every pair, between the two stages
  const size_t inv_bytes = (size_t)nunits * kUnit * kUnit * es + pair_tmp_bytes;
  const size_t max_bytes = ((size_t)nblk * 2 * sizeof(unsigned long long) + 255) / 256 * 256;   // max|inv(T_b)|, max|T_b|
  const size_t flag_bytes = ((size_t)nblk * sizeof(int) + 255) / 256 * 256 + max_bytes;
  const size_t leaf_bytes = leaf_copy ? (size_t)(left ? n : m) * kUnit * es : 0;   // one leaf's result
  const size_t need = inv_bytes + flag_bytes + leaf_bytes;
  if (!sc.done) TR_TRY(hipEventCreateWithFlags(&sc.done, hipEventDisableTiming));
  if (need > sc.cap) {
    if (sc.buf) {
      TR_TRY(hipEventSynchronize(sc.done));
      TR_TRY(hipFree(sc.buf));
      sc.buf = nullptr;
      sc.cap = 0;
    }
    TR_TRY(hipMalloc(&sc.buf, need + need / 4));
    sc.cap = need + need / 4;
  } else {
    TR_TRY(hipStreamWaitEvent(s, sc.done, 0));
  }
  const dim3 grid((unsigned)(kNB / kInvCols), (unsigned)nblk);
  const int up = upper ? 1 : 0, un = unit ? 1 : 0;
  int* flags = reinterpret_cast<int*>(static_cast<char*>(sc.buf) + inv_bytes);
  unsigned long long* gmax = reinterpret_cast<unsigned long long*>(static_cast<char*>(sc.buf) + inv_bytes + flag_bytes - max_bytes);
  TR_TRY(hipMemsetAsync(gmax, 0, max_bytes, s));
  switch (kind) {
    case F32:
      if (kNB == kLargeNB)
        hipLaunchKernelGGL((trtri_blocks<float, kLargeNB>), grid, dim3(256), 0, s, static_cast<const float*>(A), lda, (int)na, up,
                           un, static_cast<float*>(sc.buf), gmax, pairs ? 1 : 0);
      else
        hipLaunchKernelGGL((trtri_blocks<float, kSmallNB>), grid, dim3(256), 0, s, static_cast<const float*>(A), lda, (int)na, up,
                           un, static_cast<float*>(sc.buf), gmax, pairs ? 1 : 0);
      break;
    case F64:
      if (kNB == kLargeNB)
        hipLaunchKernelGGL((trtri_blocks<double, kLargeNB>), grid, dim3(256), 0, s, static_cast<const double*>(A), lda, (int)na,
                           up, un, static_cast<double*>(sc.buf), gmax, pairs ? 1 : 0);
      else
        hipLaunchKernelGGL((trtri_blocks<double, kSmallNB>), grid, dim3(256), 0, s, static_cast<const double*>(A), lda, (int)na,
                           up, un, static_cast<double*>(sc.buf), gmax, pairs ? 1 : 0);
      break;
    case C32:
      hipLaunchKernelGGL((trtri_blocks<Cplx<float>, kSmallNB>), grid, dim3(256), 0, s, static_cast<const Cplx<float>*>(A), lda,
                         (int)na, up, un, static_cast<Cplx<float>*>(sc.buf), gmax, pairs ? 1 : 0);
      break;
    default:
      hipLaunchKernelGGL((trtri_blocks<Cplx<double>, kSmallNB>), grid, dim3(256), 0, s, static_cast<const Cplx<double>*>(A), lda,
                         (int)na, up, un, static_cast<Cplx<double>*>(sc.buf), gmax, pairs ? 1 : 0);
      break;
  }
  TR_TRY(hipGetLastError());
  const double limit = growth_limit();
#define EIGAMD_DECIDE(TYPE, NBV) \
  hipLaunchKernelGGL((trtri_decide<TYPE, NBV>), dim3((unsigned)nblk), dim3(256), 0, s, static_cast<TYPE*>(sc.buf), limit, gmax, flags, pairs ? 1 : 0)
  switch (kind) {
    case F32: if (kNB == kLargeNB) EIGAMD_DECIDE(float, kLargeNB); else EIGAMD_DECIDE(float, kSmallNB); break;
    case F64: if (kNB == kLargeNB) EIGAMD_DECIDE(double, kLargeNB); else EIGAMD_DECIDE(double, kSmallNB); break;
    case C32: EIGAMD_DECIDE(Cplx<float>, kSmallNB); break;
    default:  EIGAMD_DECIDE(Cplx<double>, kSmallNB); break;
  }
#undef EIGAMD_DECIDE
  TR_TRY(hipGetLastError());
  if (pairs) {
    // The off-diagonal blocks of the pairs' inverses (pairs that have a second block: nblk / 2 of them).
    const int64_t np = nblk / 2;
    const int last_hi = (int)std::min<int64_t>(kNB, na - (2 * np - 1) * kNB);   // extent of the last pair's second block
    char* images = static_cast<char*>(sc.buf);
    char* tmp = images + (size_t)nunits * kUnit * kUnit * es;
    const int64_t s_img = kUnit * kUnit, s_tmp = kNB * kNB, s_a = 2 * kNB * (lda + 1);
    const int64_t o_hi = kNB + kNB * kUnit;                       // the second block's quadrant inside an image
    const int64_t o_off = upper ? kNB * kUnit : kNB;              // the off-diagonal quadrant that is not zero
    const char* A_off = upper ? a_at(0, kNB) : a_at(kNB, 0);      // T12 or T21 of the first pair
    const dim3 pgrid((unsigned)((kNB / 32) * (kNB / 32)), (unsigned)np);
    const int nb = (int)kNB;
#define EIGAMD_PAIR(TYPE)                                                                                                          \
    {                                                                                                                              \
      TYPE* img = reinterpret_cast<TYPE*>(images);                                                                                 \
      TYPE* t = reinterpret_cast<TYPE*>(tmp);                                                                                      \
      const TYPE* a = reinterpret_cast<const TYPE*>(A_off);                                                                        \
      if (!upper) { /* tmp = T21 inv(T11); X21 = -inv(T22) tmp */                                                                  \
        hipLaunchKernelGGL((trtri_pair_product<TYPE>), pgrid, dim3(256), 0, s, a, lda, s_a, last_hi, nb, img, kUnit, s_img, nb, nb, \
                           t, kNB, s_tmp, nb, 0, 0, 1, (const int*)nullptr);                                                                                  \
        hipLaunchKernelGGL((trtri_pair_product<TYPE>), pgrid, dim3(256), 0, s, img + o_hi, kUnit, s_img, nb, nb, t, kNB, s_tmp, nb, \
                           nb, img + o_off, kUnit, s_img, nb, 1, 1, 0, flags);                                                           \
      } else {      /* tmp = T12 inv(T22); X12 = -inv(T11) tmp */                                                                  \
        hipLaunchKernelGGL((trtri_pair_product<TYPE>), pgrid, dim3(256), 0, s, a, lda, s_a, nb, last_hi, img + o_hi, kUnit, s_img,  \
                           nb, nb, t, kNB, s_tmp, nb, 0, 0, 2, (const int*)nullptr);                                                                 \
        hipLaunchKernelGGL((trtri_pair_product<TYPE>), pgrid, dim3(256), 0, s, img, kUnit, s_img, nb, nb, t, kNB, s_tmp, nb, nb,    \
                           img + o_off, kUnit, s_img, nb, 1, 2, 0, flags);                                                              \
      }                                                                                                                            \
    }
    if (np > 0) {
      switch (kind) {
        case F32: EIGAMD_PAIR(float); break;
        case F64: EIGAMD_PAIR(double); break;
        case C32: EIGAMD_PAIR(Cplx<float>); break;
        default:  EIGAMD_PAIR(Cplx<double>); break;
      }
      TR_TRY(hipGetLastError());
    }
#undef EIGAMD_PAIR
  }
  auto inv = [&](int64_t u) { return static_cast<const char*>(sc.buf) + (size_t)u * kUnit * kUnit * es; };
  // One buffer for the pieces of every cut reduction of this solve (the solve is a chain of some dozens of products,
  // most of them of few tiles): taken from the library's pool once and given back at the end, instead of once per
  // product -- every hand-back records an event, and the marker costs the stream 5 us (dtrsm 4096: 31 of them).
  int ws_slot = -1;
  const size_t ws_bytes = std::min<size_t>((size_t)256 << 20, std::max<size_t>((size_t)4 * m * n * es, (size_t)16 * kNB * (left ? n : m) * es));
  void* ws = splitk_acquire(s, ws_bytes, &ws_slot);   // (null: no buffer to be had; the products then ask for themselves)
  struct Release {
    int slot; hipStream_t s;
    ~Release() { if (slot >= 0) splitk_release(slot, s); }
  } release_ws{ws_slot, s};

  const bool notrans = op == OP_N || op == OP_CONJ;
  const bool eff_lower = (!upper) == notrans;  // op(A) is lower triangular
  // The substitution form of one diagonal block (a no-op unless trtri_decide flagged it).
  const int s_trans = notrans ? 0 : 1, s_conj = (op == OP_C || op == OP_CONJ) ? 1 : 0, s_right = left ? 0 : 1;
  const int s_rev = (left ? eff_lower : !eff_lower) ? 0 : 1;  // the canonical matrix (op(A), or op(A)^T on the right) is upper
  auto subst = [&](int64_t j, int64_t p0, int64_t pb) -> hipError_t {
    const int64_t nrhs = left ? n : m;
    const dim3 grid((unsigned)((nrhs + 255) / 256));
    const void* Ajj = a_at(p0, p0);
    void* Bj = left ? b_at(p0, 0) : b_at(0, p0);
    const int64_t inc_i = left ? 1 : ldb, inc_c = left ? ldb : 1;
#define EIGAMD_SUBST(TYPE)                                                                                                      \
    hipLaunchKernelGGL((trsm_block_subst<TYPE>), grid, dim3(256), 0, s, flags + j, static_cast<const TYPE*>(Ajj), lda, (int)pb,  \
                       s_rev, s_right, s_trans, s_conj, un, static_cast<TYPE*>(Bj), inc_i, inc_c, (int)nrhs)
    switch (kind) {
      case F32: EIGAMD_SUBST(float); break;
      case F64: EIGAMD_SUBST(double); break;
      case C32: EIGAMD_SUBST(Cplx<float>); break;
      default:  EIGAMD_SUBST(Cplx<double>); break;
    }
#undef EIGAMD_SUBST
    return hipGetLastError();
  };
  // Recursive halving over the diagonal blocks [j0, j1): solve the half that does not depend
  // on the other, update the other half with ONE product whose reduction length is the size
  // of the first half (so the large updates run at GEMM speed and B is re-read O(log) times),
  // solve the other half.  Left: blocks are row blocks of B, forward when op(A) is lower.
  // Right (X op(A) = B): blocks are column blocks of B, forward when op(A) is upper.
  struct Rec {
    std::function<hipError_t(int64_t, int64_t)> run;
  } rec;
  const int64_t ext = na;
  rec.run = [&](int64_t j0, int64_t j1) -> hipError_t {
    if (j1 - j0 == 1) {
      const int64_t p0 = j0 * kUnit, pb = std::min<int64_t>(kUnit, ext - p0);
      // the substitution form of the whole unit (one flag per pair: trtri_decide), a no-op unless flagged
      TR_TRY(subst(unit_blocks * j0, p0, pb));
      if (leaf_copy) {  // the product into the scratch image (any tile may run it), then back
        // (Where the launcher cuts the leaf's reduction into pieces -- it does for most leaves: few tiles -- the pass
        // over the pieces writes B_j itself and there is nothing to copy: 16 us less per leaf, copy and the two gaps
        // around it.)
        char* leaf = static_cast<char*>(sc.buf) + inv_bytes + flag_bytes;
        bool direct = false;
        if (left) {
          TR_TRY(gemm(kind, op, OP_N, pb, n, pb, 1.0, inv(j0), kUnit, b_at(p0, 0), ldb, 0.0, leaf, pb, s, ws, ws ? ws_bytes : 0, b_at(p0, 0), ldb, &direct));
          if (direct) return hipSuccess;
          return hipMemcpy2DAsync(b_at(p0, 0), (size_t)ldb * es, leaf, (size_t)pb * es, (size_t)pb * es, (size_t)n,
                                  hipMemcpyDeviceToDevice, s);
        }
        TR_TRY(gemm(kind, OP_N, op, m, pb, pb, 1.0, b_at(0, p0), ldb, inv(j0), kUnit, 0.0, leaf, m, s, ws, ws ? ws_bytes : 0, b_at(0, p0), ldb, &direct));
        if (direct) return hipSuccess;
        return hipMemcpy2DAsync(b_at(0, p0), (size_t)ldb * es, leaf, (size_t)m * es, (size_t)m * es, (size_t)pb,
                                hipMemcpyDeviceToDevice, s);
      }
      // (in place: leaves of one block, kUnit == kNB)
      if (left)  // B_j <- op(inv_j) * B_j, in place (one tile row)
        return gemm(kind, op, OP_N, pb, n, pb, 1.0, inv(j0), kUnit, b_at(p0, 0), ldb, 0.0, b_at(p0, 0), ldb, s, ws, ws ? ws_bytes : 0);
      // B_j <- B_j * op(inv_j), in place (one tile column)
      return gemm(kind, OP_N, op, m, pb, pb, 1.0, b_at(0, p0), ldb, inv(j0), kUnit, 0.0, b_at(0, p0), ldb, s, ws, ws ? ws_bytes : 0);
    }
    const int64_t jm = j0 + (j1 - j0 + 1) / 2;
    const int64_t lo0 = j0 * kUnit, lo1 = jm * kUnit, hi1 = std::min<int64_t>(j1 * kUnit, ext);
    const int64_t nlo = lo1 - lo0, nhi = hi1 - lo1;
    const bool low_first = left ? eff_lower : !eff_lower;   // which half is solved first
    TR_TRY(low_first ? rec.run(j0, jm) : rec.run(jm, j1));
    // f = the finished half, q = the half still to solve
    const int64_t f0 = low_first ? lo0 : lo1, fn = low_first ? nlo : nhi;
    const int64_t q0 = low_first ? lo1 : lo0, qn = low_first ? nhi : nlo;
    if (left) {
      // B_q -= op(A)(q, f) * B_f
      const void* Aqf = notrans ? a_at(q0, f0) : a_at(f0, q0);
      TR_TRY(gemm(kind, op, OP_N, qn, n, fn, -1.0, Aqf, lda, b_at(f0, 0), ldb, 1.0, b_at(q0, 0), ldb, s, ws, ws ? ws_bytes : 0));
    } else {
      // B_q -= B_f * op(A)(f, q)
      const void* Afq = notrans ? a_at(f0, q0) : a_at(q0, f0);
      TR_TRY(gemm(kind, OP_N, op, m, qn, fn, -1.0, b_at(0, f0), ldb, Afq, lda, 1.0, b_at(0, q0), ldb, s, ws, ws ? ws_bytes : 0));
    }
    return low_first ? rec.run(jm, j1) : rec.run(j0, jm);
  };
  TR_TRY(rec.run(0, nunits));
  TR_TRY(hipEventRecord(sc.done, s));
  if (!(alpha[0] == 1.0 && alpha[1] == 0.0)) {  // B *= alpha (level3_impl.h:174-175)
    GemmCall c;
    c.kind = kind;
    c.opa = OP_N; c.opb = OP_N;
    c.m = m; c.n = n; c.k = 0;
    c.alpha[0] = 0.0; c.alpha[1] = 0.0;
    c.beta[0] = alpha[0]; c.beta[1] = alpha[1];
    c.A = nullptr; c.lda = 1;
    c.B = nullptr; c.ldb = 1;
    c.C = B; c.ldc = ldb;
    switch (kind) {
      case F32: TR_TRY(launch_gemm_f32(c, s)); break;
      case F64: TR_TRY(launch_gemm_f64(c, s)); break;
      case C32: TR_TRY(launch_gemm_c32(c, s)); break;
      default:  TR_TRY(launch_gemm_c64(c, s)); break;
    }
  }
  return hipSuccess;
}

}  // namespace eigamd
