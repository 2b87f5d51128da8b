// Device-resident blocked LU with partial pivoting, right-looking, whose trailing
// update A22 -= A21 * A12 runs on the MFMA GEMM kernels.
//
// Mirrors partial_lu_impl::blocked_lu / unblocked_lu of the reference
// (Eigen/src/LU/PartialPivLU.h:357-498): same block-size rule (:443-448), the same
// two-level structure (a panel of blockSize columns is itself factorised with
// blocks of 16 and an unblocked base case, :472-473), the same pivot choice (first
// row of largest |a|, :375-377), zero-pivot bookkeeping (:391-396) and the same
// order of row transpositions on the left (A_0) and right (A_2) parts (:479-489).
// Everything stays in HBM and the pivot indices never visit the host.  The unblocked
// base case (a panel of at most 16 columns) is ONE kernel: every thread keeps one row
// of the panel in registers, the workgroups agree on each pivot through a small
// exchange area and a grid-wide barrier per column (lu_base_fused).  Panels too tall
// for all their workgroups to be resident at once fall back to a chain of small
// kernels per column (lu_colmax / lu_pivot / lu_update), where the kernel boundary is
// the grid-wide synchronisation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "gemm_launch.hpp"
#include "lu.hpp"
#include "trsm.hpp"

namespace eigamd {

namespace {

#ifndef EIGAMD_LU_ROWS_PER_BLOCK
#define EIGAMD_LU_ROWS_PER_BLOCK 256
#endif
#ifndef EIGAMD_LU_POLL_DELAY
#define EIGAMD_LU_POLL_DELAY 0  // in units of 64 clocks
#endif
constexpr int kRowsPerBlock = EIGAMD_LU_ROWS_PER_BLOCK;  // rows of the panel one workgroup owns
constexpr int kMaxBase = 16;        // widest unblocked panel (UnBlockedBound, PartialPivLU.h:336)

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant in
// every iteration, whatever the optimiser thinks of the body's size.
template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// The scalar of a factorisation is float, double or a complex number of either (the reference's blocked_lu is a
// template over the scalar and is bound for all four, PartialPivLU_LAPACKE.h:74-79).  Cx<R> is the (re, im) pair as
// the callers store it, with the arithmetic the kernels need; division is Smith's, which does not overflow where
// |b|^2 would.  Everything below is written once over the scalar T; what is always real -- the size of a pivot
// candidate (scalar_score_coeff_op, PartialPivLU.h:375), the growth of an inverse -- has the type Real<T>.
template <typename R> struct alignas(2 * sizeof(R)) Cx {
  R x, y;
  Cx() = default;
  __host__ __device__ constexpr Cx(R re) : x(re), y(0) {}
  __host__ __device__ constexpr Cx(R re, R im) : x(re), y(im) {}
};
template <typename R> __device__ __forceinline__ Cx<R> operator+(Cx<R> a, Cx<R> b) { return {a.x + b.x, a.y + b.y}; }
template <typename R> __device__ __forceinline__ Cx<R> operator-(Cx<R> a, Cx<R> b) { return {a.x - b.x, a.y - b.y}; }
template <typename R> __device__ __forceinline__ Cx<R> operator-(Cx<R> a) { return {-a.x, -a.y}; }
template <typename R> __device__ __forceinline__ Cx<R> operator*(Cx<R> a, Cx<R> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename R> __device__ __forceinline__ Cx<R> operator/(Cx<R> a, Cx<R> b) {
  if (fabs(b.x) >= fabs(b.y)) {
    const R r = b.y / b.x, d = b.x + b.y * r;
    return {(a.x + a.y * r) / d, (a.y - a.x * r) / d};
  }
  const R r = b.x / b.y, d = b.x * r + b.y;
  return {(a.x * r + a.y) / d, (a.y * r - a.x) / d};
}
template <typename R> __device__ __forceinline__ Cx<R>& operator+=(Cx<R>& a, Cx<R> b) { a.x += b.x; a.y += b.y; return a; }
template <typename R> __device__ __forceinline__ Cx<R>& operator-=(Cx<R>& a, Cx<R> b) { a.x -= b.x; a.y -= b.y; return a; }
template <typename T> struct RealOf { using type = T; };
template <typename R> struct RealOf<Cx<R>> { using type = R; };
template <typename T> using Real = typename RealOf<T>::type;
__device__ __forceinline__ float absval(float v) { return fabs(v); }
__device__ __forceinline__ double absval(double v) { return fabs(v); }
template <typename R> __device__ __forceinline__ R absval(Cx<R> v) { return hypot(v.x, v.y); }

template <typename R> struct Candidate {
  R val;    // |a|, or -1 for "no row"
  int row;  // row index relative to the sub-panel
};

template <typename T> struct Side {
  T u[kMaxBase];  // pivot row of the sub-panel after the swap
  T pivot;
  int zero;       // the pivot is exactly zero: skip the scaling
};

template <typename T> __device__ __forceinline__ bool better(T v, int r, T ov, int orow) {
  return (v > ov) | ((v == ov) & (r < orow));
}

// Data-parallel-primitive moves: what lane `ctrl` designates holds, or this lane's own value
// where the control designates no lane (or the row mask leaves this lane's row out).
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_take(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_take(float v) {
  return __int_as_float(dpp_take<CTRL, ROW_MASK>(__float_as_int(v)));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_take(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)dpp_take<CTRL, ROW_MASK>((int)(unsigned)b);
  const unsigned hi = (unsigned)dpp_take<CTRL, ROW_MASK>((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Best candidate of a wavefront, left in lane 63: shifts by 1, 2, 4, 8 inside the rows of 16
// lanes (row_shr), then lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast15), then lane 31 into
// rows 2 and 3 (row_bcast31).  Register moves, against a trip through the LDS crossbar per shuffle.
template <typename T> __device__ __forceinline__ void wave_best_to_lane63(T& val, int& row) {
#define EIGAMD_BEST_STEP(CTRL, MASK)                                   \
  {                                                                    \
    const T ov = dpp_take<CTRL, MASK>(val);                            \
    const int orow = dpp_take<CTRL, MASK>(row);                        \
    const bool take = better(ov, orow, val, row);                      \
    val = take ? ov : val;                                             \
    row = take ? orow : row;                                           \
  }
  EIGAMD_BEST_STEP(0x111, 0xf)  // row_shr:1
  EIGAMD_BEST_STEP(0x112, 0xf)  // row_shr:2
  EIGAMD_BEST_STEP(0x114, 0xf)  // row_shr:4
  EIGAMD_BEST_STEP(0x118, 0xf)  // row_shr:8
  EIGAMD_BEST_STEP(0x142, 0xa)  // row_bcast15 into rows 1, 3
  EIGAMD_BEST_STEP(0x143, 0xc)  // row_bcast31 into rows 2, 3
#undef EIGAMD_BEST_STEP
}
__device__ __forceinline__ int from_lane63(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ float from_lane63(float v) { return __int_as_float(from_lane63(__float_as_int(v))); }
__device__ __forceinline__ double from_lane63(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)from_lane63((int)(unsigned)b);
  const unsigned hi = (unsigned)from_lane63((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Largest |a| with the smallest row index among ties, over one workgroup.
template <typename T> __device__ Candidate<T> block_best(T val, int row) {
  __shared__ T s_val[kRowsPerBlock / 64];
  __shared__ int s_row[kRowsPerBlock / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T ov = __shfl_down(val, off, 64);
    const int orow = __shfl_down(row, off, 64);
    if (better(ov, orow, val, row)) { val = ov; row = orow; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { s_val[wave] = val; s_row[wave] = row; }
  __syncthreads();
  Candidate<T> c = {s_val[0], s_row[0]};
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
    if (better(s_val[w], s_row[w], c.val, c.row)) { c.val = s_val[w]; c.row = s_row[w]; }
  return c;
}

// partials[b] = best candidate of column `col` over rows [row_begin, rows) of block b.
template <typename T>
__global__ void lu_colmax(const T* __restrict__ P, int64_t ld, int row_begin, int rows, int col,
                          Candidate<Real<T>>* __restrict__ partials) {
  const int r = row_begin + blockIdx.x * kRowsPerBlock + threadIdx.x;
  Real<T> val = (Real<T>)-1;
  if (r < rows) val = absval(P[r + (int64_t)col * ld]);
  const Candidate<Real<T>> c = block_best<Real<T>>(val, r);
  if (threadIdx.x == 0) partials[blockIdx.x] = c;
}

// One wavefront: pick the pivot of column jj from the per-block candidates, record
// the transposition, swap the two rows across the sub-panel, publish the pivot row.
template <typename T>
__global__ void lu_pivot(T* __restrict__ P, int64_t ld, int jbw, int jj, const Candidate<Real<T>>* __restrict__ partials,
                         int nparts, int* __restrict__ transp, int row_offset, int col_global,
                         Side<T>* __restrict__ side, int* __restrict__ first_zero, int* __restrict__ nswaps) {
  using R = Real<T>;
  const int lane = threadIdx.x;
  R val = (R)-1;
  int row = 0x7fffffff;
  for (int i = lane; i < nparts; i += 64) {
    const Candidate<R> c = partials[i];
    if (better(c.val, c.row, val, row)) { val = c.val; row = c.row; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const R ov = __shfl_down(val, off, 64);
    const int orow = __shfl_down(row, off, 64);
    if (better(ov, orow, val, row)) { val = ov; row = orow; }
  }
  val = __shfl(val, 0, 64);
  row = __shfl(row, 0, 64);
  if (row == 0x7fffffff) {  // nothing comparable in the column (NaNs): no exchange, pivot = the diagonal entry
    row = jj;
    val = (R)1;
  }
  const bool zero = !(val != (R)0);  // exactly zero
  if (lane == 0) {
    transp[0] = row_offset + row;
    if (zero) {
      atomicCAS(first_zero, -1, col_global);  // keep the FIRST zero pivot (PartialPivLU.h:391-396)
    } else if (row != jj) {
      atomicAdd(nswaps, 1);
    }
  }
  if (lane < jbw) {
    T top = P[jj + (int64_t)lane * ld];
    if (!zero && row != jj) {
      const T other = P[row + (int64_t)lane * ld];
      P[row + (int64_t)lane * ld] = top;
      P[jj + (int64_t)lane * ld] = other;
      top = other;
    }
    side->u[lane] = top;
    if (lane == jj) { side->pivot = top; side->zero = zero ? 1 : 0; }
  }
}

// Rows below the pivot: l = a / pivot, rank-1 update of the remaining columns of the
// sub-panel, and the candidate for the next column's pivot.
template <typename T>
__global__ void lu_update(T* __restrict__ P, int64_t ld, int rows, int jbw, int jj, const Side<T>* __restrict__ side,
                          Candidate<Real<T>>* __restrict__ partials) {
  __shared__ T s_u[kMaxBase];
  __shared__ T s_pivot;
  __shared__ int s_zero;
  if (threadIdx.x < kMaxBase) s_u[threadIdx.x] = side->u[threadIdx.x];
  if (threadIdx.x == 0) { s_pivot = side->pivot; s_zero = side->zero; }
  __syncthreads();
  const int r = jj + 1 + blockIdx.x * kRowsPerBlock + threadIdx.x;
  Real<T> cand = (Real<T>)-1;
  if (r < rows) {
    T l = P[r + (int64_t)jj * ld];
    if (!s_zero) {
      l = l / s_pivot;  // the reference divides (lu.col(k).tail /= pivot, :389)
      P[r + (int64_t)jj * ld] = l;
    }
    for (int c = jj + 1; c < jbw; ++c) {
      T* dst = P + r + (int64_t)c * ld;
      const T x = *dst - l * s_u[c];
      *dst = x;
      if (c == jj + 1) cand = absval(x);
    }
  }
  if (jj + 1 < jbw) {
    const Candidate<Real<T>> c = block_best<Real<T>>(cand, r);
    if (threadIdx.x == 0) partials[blockIdx.x] = c;
  }
}

// max|x| over the entries a kernel of the inversion writes, folded into one word per inverse: the
// bit pattern of a non-negative double orders like the number itself (NaN above infinity), so an
// integer atomicMax does it.  lu_inverse_finish reads and resets the word.
__device__ __forceinline__ void note_inverse_max(unsigned long long* gmax, double local_abs_max) {
  unsigned long long b = (unsigned long long)__double_as_longlong(fabs(local_abs_max));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_down(b, off, 64);
    b = o > b ? o : b;
  }
  if ((threadIdx.x & 63) == 0 && b != 0) atomicMax(gmax, b);
}
__device__ __forceinline__ double abs_max2(double m, double v) {
  const double a = fabs(v);
  return (a > m || a != a) ? a : m;
}

// ---- the inverse of a panel's unit lower triangle, built while the panel is factorised ---------------------
//
// The solve A12 <- inv(L11) A12 behind a panel is one product with the explicit inverse X of the panel's bs x bs
// unit lower triangle (make_inverse), and X used to be worked out by a chain of five launches between the panel
// and everything that waits for it (0.11 ms of a 1.6 ms step at 4000 rows).  X can be built row block by row
// block instead, as the panel's 16-column blocks are finished: rows [i0, i1) of X need rows [0, i1) of L only,
//     X(i, :) = e_i - sum_{k < i} L(i, k) X(k, :),
// and those rows of L are final once block i / 16 has been factorised and its transpositions have reached the
// columns to its left.  A few extra workgroups of the next block's base-case launch do it (that launch spends 50 us
// and more on its sixteen exchanges, the rows take less); what remains behind the panel is the last 16 rows and the
// look at max|X| (lu_inverse_rows, lu_inverse_tail: two short launches).
constexpr int kPlanSpan = 256;
constexpr int kMaxSub = 4;  // an outer step of the look-ahead spans at most kMaxSub * kPlanSpan columns
struct SwapPlan {   // (see lu_swap_plan)
  int count;
  int dst[2 * kPlanSpan];
  int src[2 * kPlanSpan];
};
template <typename T> struct InvJob {
  const T* L = nullptr;   // the bs x bs block (unit lower triangle), leading dimension ldl
  int64_t ldl = 0;
  T* X = nullptr;         // bs x bs, leading dimension bs
  T* XT = nullptr;        // the transpose of X (scratch of the job: what it reads back is contiguous along its lanes)
  unsigned long long* gmax = nullptr;
  int bs = 0, i0 = 0, i1 = 0;   // rows [i0, i1) are to be formed; rows [0, i0) are there
  // (the launch behind the panel only) the panel's transpositions [k0, k1) as one permutation, by one more workgroup
  SwapPlan* plan = nullptr;
  const int* transp = nullptr;
  int k0 = 0, k1 = 0;
};
// A job is dealt over kInvGroups workgroups of 256 threads, 64 columns of X each: lane = column, wavefront = four of
// the sixteen rows of a row block.  With M = E - L(rows, above) X(above, :) the rows are X(rows, :) = inv(L(rows, rows)) M:
// the sums over the rows above are independent of each other, the 16 x 16 triangle is inverted once by 16 threads.
constexpr int kInvGroups = 4, kInvChunk = 64;
template <typename T> struct InvShared {
  T l[kInvChunk][kMaxBase + 1];    // l[kk][i] = L(r0 + i, k0 + kk); afterwards m[j][lane] in the same memory
  T d[kMaxBase][kMaxBase + 1];     // d[kk][i] = L(r0 + i, r0 + kk), i > kk
  T inv[kMaxBase][kMaxBase + 1];   // inv[j][i] = inv(L(rows, rows))(i, j)
};
static_assert(kInvChunk * (kMaxBase + 1) >= kMaxBase * 65, "m (16 x 65) lives where l was");
static_assert(kMaxBase * 4 <= 64, "a wavefront publishes a row of the widest scalar with one store per lane");

template <typename T, int BATCH = 8> __device__ __forceinline__ void inverse_rows_job(const InvJob<T> job, const int group, InvShared<T>& sh) {
  const int t = threadIdx.x, cl = t & 63, rg = t >> 6;
  const int bs = job.bs;
  const int c0 = group * 64, c = c0 + cl;
  const T* __restrict__ L = job.L;
  const int64_t ldl = job.ldl;
  T* m = &sh.l[0][0];   // m[j * 65 + lane]
  double mx = 0.0;
  if (c0 < bs) {   // (uniform: a group beyond the block has nothing to do)
  for (int r0 = job.i0; r0 < job.i1; r0 += kMaxBase) {
    const int nr = min(kMaxBase, job.i1 - r0);
    T acc[4] = {(T)0, (T)0, (T)0, (T)0};
    // the rows above: X(k, c) = 0 for k < c, so k starts at this group's first column
    for (int k0 = c0; k0 < r0; k0 += kInvChunk) {
      __syncthreads();
      for (int idx = t; idx < kInvChunk * kMaxBase; idx += 256) {
        const int i = idx % kMaxBase, kk = idx / kMaxBase;
        sh.l[kk][i] = (i < nr && k0 + kk < r0) ? L[(r0 + i) + (int64_t)(k0 + kk) * ldl] : (T)0;
      }
      __syncthreads();
      if (c < bs) {
        // BATCH loads in flight per thread.  (8 where the job shares its kernel, and that kernel's register budget, with a
        // base case; the launch behind the panel, which stands between the panel and the next one, takes 32.)
#pragma unroll 1
        for (int h = 0; h < kInvChunk; h += BATCH) {
          T xk[BATCH];
#pragma unroll
          for (int u = 0; u < BATCH; ++u) xk[u] = k0 + h + u < r0 ? job.XT[c + (int64_t)(k0 + h + u) * bs] : (T)0;
#pragma unroll
          for (int u = 0; u < BATCH; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += sh.l[h + u][4 * rg + q] * xk[u];
        }
      }
    }
    __syncthreads();
    // the triangle on the diagonal and its inverse (thread j: column j, by substitution)
    for (int idx = t; idx < kMaxBase * kMaxBase; idx += 256) {
      const int i = idx % kMaxBase, kk = idx / kMaxBase;
      sh.d[kk][i] = (i < nr && kk < i) ? L[(r0 + i) + (int64_t)(r0 + kk) * ldl] : (T)0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) m[(4 * rg + q) * 65 + cl] = ((r0 + 4 * rg + q == c) ? (T)1 : (T)0) - acc[q];
    __syncthreads();
    if (t < kMaxBase) {   // column t of the triangle's inverse, kept in LDS as it grows (no register array: see above)
#pragma unroll 1
      for (int i = 0; i < kMaxBase; ++i) {
        T v = i == t ? (T)1 : (T)0;
#pragma unroll 1
        for (int kk = t; kk < i; ++kk) v -= sh.d[kk][i] * sh.inv[t][kk];
        sh.inv[t][i] = i < t ? (T)0 : v;
      }
    }
    __syncthreads();
    if (c < bs) {
      T v[4] = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll 1
      for (int j = 0; j < kMaxBase; ++j) {   // (inv[j][i] = 0 for i < j)
        const T mj = m[j * 65 + cl];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += sh.inv[j][4 * rg + q] * mj;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 4 * rg + q;
        if (i < nr) {
          job.X[(r0 + i) + (int64_t)c * bs] = v[q];
          job.XT[c + (int64_t)(r0 + i) * bs] = v[q];
          mx = abs_max2(mx, (double)absval(v[q]));
        }
      }
    }
    __syncthreads();   // (the next row block reads these rows of XT back: same workgroup)
  }
  }
  note_inverse_max(job.gmax, mx);
}

// Columns of C a workgroup owns and how many of them a thread loads together.  32 / 16 against
// 64 / 8: LU 8192^2 58.2 -> 55.9 ms (the kernel waits on its loads of C; more of them in flight and
// twice the workgroups).  32 / 32 changes the float result in the last place (the compiler
// reassociates something) and with it a pivot of the parity test: not used.
#ifndef EIGAMD_LU_RANK_GROUP
#define EIGAMD_LU_RANK_GROUP 16
#endif
#ifndef EIGAMD_LU_RANK_COLS
#define EIGAMD_LU_RANK_COLS 32
#endif
constexpr int kRankCols = EIGAMD_LU_RANK_COLS;
constexpr int kRankCounters = 8;   // arrival counters of lu_rowops_rank_update: bytes [208, 240) of the scratch header
// lu_rowops and lu_rank_update of one 16-column block in ONE launch (a launch of the panel's chain costs 10 us and
// more whatever it does; at 8192 rows the two were 14 + 13 us of a block's 95).  What the row operations change are
// the block's "special" rows -- its jb top rows and the rows its transpositions name -- in the columns right of the
// block; what the rank update needs of them is U12, the top rows after exchanges and forward substitution.  Every
// workgroup of a column block reads the special rows of its 32 columns as they are, works out U12 and the exchanged
// rows in LDS (redundantly: 32 rows x 32 columns), and updates its own 256 rows, skipping special ones.  Nobody
// writes a special row while anybody may still read it: a workgroup counts itself in once its reads have landed in
// LDS, and the workgroup that completes the count of its column block writes that block's special rows at the end --
// U12 into the top rows, and "old top row - L(p, :) U12" into every named row p -- and puts the counter back to
// zero.  No workgroup waits for another.  The columns left of the block inside the panel only exchange rows: extra
// workgroups at the end of the grid, one thread per column.  Same operations per element, in the same order, as
// lu_rowops followed by lu_rank_update.  Row and column arguments are absolute (`base` is the matrix).
// (As a device function, so that the part of a block's update that nothing waits for -- the columns beyond the next
// block's -- can run as extra workgroups of the next block's base-case launch, beside its sixteen exchanges.)
template <typename T> struct RowopsArgs {
  T* base = nullptr;
  int64_t ld = 0;
  const int* transp = nullptr;
  int k0 = 0, jb = 0, cL = 0, cR = 0, nright = 0, cLeft0 = 0, nleft = 0, trows = 0, nrb = 0, ncb = 0;
  int* counters = nullptr;
  __host__ __device__ int workgroups() const { return nrb * ncb + (nleft + 255) / 256; }
};
template <typename T> struct RowopsShared {
  T rows[2 * kMaxBase][kRankCols + 1];  // [0, 16): the top rows, then U12; [16, 32): the rows named by the transpositions
  T l[kMaxBase * kMaxBase];
  int p[kMaxBase];
  int slot[kMaxBase];   // where row p[i] lives in rows
  int last;
};
// EARLY (the stand-alone launch, which stands between two base cases): a thread's own row of L and its first group of C
// are asked for at the very start, so that they travel while the special rows are read and solved -- two dependent trips
// to memory instead of three.  (Not where the block rides in a base case's kernel: the registers are that kernel's.)
template <typename T, int GROUP = EIGAMD_LU_RANK_GROUP, bool EARLY = false>
__device__ __forceinline__ void rowops_rank_update_block(const RowopsArgs<T> args, const int b, RowopsShared<T>& sh) {
  T* __restrict__ base = args.base;
  const int64_t ld = args.ld;
  const int* __restrict__ transp = args.transp;
  const int k0 = args.k0, jb = args.jb, cL = args.cL, cR = args.cR, nright = args.nright, cLeft0 = args.cLeft0, nleft = args.nleft;
  const int trows = args.trows, nrb = args.nrb, ncb = args.ncb;
  int* __restrict__ counters = args.counters;
  auto& s_rows = sh.rows;
  auto& s_l = sh.l;
  auto& s_p = sh.p;
  auto& s_slot = sh.slot;
  int& s_last = sh.last;
  const int t = threadIdx.x;
  T a_early[EARLY ? kMaxBase : 1], c_early[EARLY ? GROUP : 1];
  if constexpr (EARLY) {
    if (b < nrb * ncb) {
      const int rb_ = b % nrb, cb_ = b / nrb;
      const int lr_ = rb_ * 256 + t, nc_ = min(kRankCols, nright - cb_ * kRankCols);
      const bool live = lr_ < trows;   // (a special row's loads are harmless: it is skipped below)
      const T* Lr = base + (k0 + jb + lr_) + (int64_t)cL * ld;
      const T* Cr_ = base + (k0 + jb + lr_) + (int64_t)(cR + cb_ * kRankCols) * ld;
#pragma unroll
      for (int k = 0; k < kMaxBase; ++k) a_early[k] = (live && k < jb) ? Lr[(int64_t)k * ld] : (T)0;
#pragma unroll
      for (int u = 0; u < GROUP; ++u) c_early[u] = (live && u < nc_) ? Cr_[(int64_t)u * ld] : (T)0;
    }
  }
  if (t < kMaxBase) s_p[t] = t < jb ? transp[k0 + t] : -1;
  for (int i = t; i < jb * jb; i += 256) s_l[(i % jb) + (i / jb) * kMaxBase] = base[(k0 + i % jb) + (int64_t)(cL + i / jb) * ld];
  __syncthreads();
  if (b >= nrb * ncb) {  // the columns to the left: exchanges only
    const int c = (b - nrb * ncb) * 256 + t;
    if (c < nleft) {
      T* col = base + (int64_t)(cLeft0 + c) * ld;
      for (int i = 0; i < jb; ++i) {
        const int p = s_p[i];
        if (p != k0 + i) {
          const T v = col[k0 + i];
          col[k0 + i] = col[p];
          col[p] = v;
        }
      }
    }
    return;
  }
  const int rb = b % nrb, cb = b / nrb;
  const int c0 = cb * kRankCols;
  const int nc = min(kRankCols, nright - c0);
  if (t < kMaxBase) {
    int slot = -1;
    if (t < jb) {
      const int p = s_p[t];
      if (p < k0 + jb) {
        slot = p - k0;            // one of the top rows
      } else {
        slot = kMaxBase + t;      // a row below them: the slot of the first transposition that names it
        for (int j = t - 1; j >= 0; --j)
          if (s_p[j] == p) slot = kMaxBase + j;
      }
    }
    s_slot[t] = slot;
  }
  __syncthreads();
  T* colsR = base + (int64_t)(cR + c0) * ld;
  for (int idx = t; idx < 2 * kMaxBase * kRankCols; idx += 256) {
    const int half = idx / (kMaxBase * kRankCols), e = idx % (kMaxBase * kRankCols);
    const int i = e % kMaxBase, j = e / kMaxBase;   // consecutive threads: consecutive top rows of one column
    T v = (T)0;
    if (i < jb && j < nc) {
      if (half == 0) v = colsR[(k0 + i) + (int64_t)j * ld];
      else if (s_slot[i] == kMaxBase + i) v = colsR[s_p[i] + (int64_t)j * ld];
    }
    s_rows[half * kMaxBase + i][j] = v;
  }
  __syncthreads();
  // every special row this workgroup reads is in LDS now
  if (t == 0) {
    const int arrived = __hip_atomic_fetch_add(counters + cb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = arrived == nrb - 1;
    if (last) __hip_atomic_store(counters + cb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last;
  }
  if (t < kRankCols) {  // one thread per column: the exchanges in order, then the forward substitution (as lu_rowops)
    for (int i = 0; i < jb; ++i) {
      if (s_p[i] != k0 + i) {
        const int sl = s_slot[i];
        const T v = s_rows[i][t];
        s_rows[i][t] = s_rows[sl][t];
        s_rows[sl][t] = v;
      }
    }
    T x[kMaxBase];
#pragma unroll
    for (int i = 0; i < kMaxBase; ++i) x[i] = i < jb ? s_rows[i][t] : (T)0;
#pragma unroll
    for (int i = 0; i < kMaxBase; ++i) {
      if (i < jb) {
#pragma unroll
        for (int rr = i + 1; rr < kMaxBase; ++rr)
          if (rr < jb) x[rr] -= s_l[rr + i * kMaxBase] * x[i];
      }
    }
#pragma unroll
    for (int i = 0; i < kMaxBase; ++i)
      if (i < jb) s_rows[i][t] = x[i];
  }
  __syncthreads();
  const int lr = rb * 256 + t;        // row of the trailing part
  const int r = k0 + jb + lr;         // the same, absolute
  bool plain = lr < trows;
  if (plain)
    for (int i = 0; i < jb; ++i)
      if (s_p[i] == r) plain = false;  // a special row: the workgroup that completes the count writes it
  if (plain) {
    T a[kMaxBase];
#pragma unroll
    for (int k = 0; k < kMaxBase; ++k) {
      if constexpr (EARLY) a[k] = a_early[k];
      else a[k] = k < jb ? base[r + (int64_t)(cL + k) * ld] : (T)0;
    }
    T* Cr = colsR + r;
    int j0 = 0;
    constexpr int kGroup = GROUP;   // (8 where the block shares a base case's kernel and its register budget)
    for (; j0 + kGroup <= nc; j0 += kGroup) {
      T c[kGroup];
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        if (EARLY && j0 == 0) c[u] = c_early[u];
        else c[u] = Cr[(int64_t)(j0 + u) * ld];
      }
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        T acc = (T)0;
#pragma unroll
        for (int k = 0; k < kMaxBase; ++k) acc += a[k] * s_rows[k][j0 + u];
        c[u] -= acc;
      }
#pragma unroll
      for (int u = 0; u < kGroup; ++u) Cr[(int64_t)(j0 + u) * ld] = c[u];
    }
    for (; j0 < nc; ++j0) {
      T acc = (T)0;
#pragma unroll
      for (int k = 0; k < kMaxBase; ++k) acc += a[k] * s_rows[k][j0];
      Cr[(int64_t)j0 * ld] -= acc;
    }
  }
  if (s_last) {  // every workgroup of this column block has read the special rows: they may be written now
    for (int e = t; e < kMaxBase * kRankCols; e += 256) {
      const int i = e % kMaxBase, j = e / kMaxBase;
      if (i < jb && j < nc) colsR[(k0 + i) + (int64_t)j * ld] = s_rows[i][j];
    }
    for (int e = t; e < kMaxBase * kRankCols; e += 256) {
      const int i = e % kMaxBase, j = e / kMaxBase;
      if (i < jb && j < nc && s_slot[i] == kMaxBase + i) {
        const int p = s_p[i];
        T acc = (T)0;
#pragma unroll
        for (int k = 0; k < kMaxBase; ++k) acc += (k < jb ? base[p + (int64_t)(cL + k) * ld] : (T)0) * s_rows[k][j];
        colsR[p + (int64_t)j * ld] = s_rows[kMaxBase + i][j] - acc;
      }
    }
  }
}

// What the launch of a block's base case carries besides the base case: rows of the enclosing panel's inverse and
// the part of the block before's update that nothing waits for.
template <typename T> struct Ride {
  InvJob<T> inv;
  RowopsArgs<T> far;
  __host__ __device__ int inv_groups() const { return inv.i1 > inv.i0 ? kInvGroups : 0; }
  __host__ __device__ int far_groups() const { return far.nright > 0 ? far.workgroups() : 0; }
};
// The extra workgroups of a base-case launch: `extra` = index behind the base case's own workgroups.
template <typename T> __device__ __forceinline__ void run_ride(const Ride<T> ride, const int extra) {
  if (extra < ride.inv_groups()) {
    __shared__ InvShared<T> s_inv;
    inverse_rows_job<T>(ride.inv, extra, s_inv);
    return;
  }
  __shared__ RowopsShared<T> s_far;
  rowops_rank_update_block<T, 8>(ride.far, extra - ride.inv_groups(), s_far);
}

// ---- the base case in one kernel ---------------------------------------------------
//
// The workgroups of the fused base case agree on each pivot through an exchange area in
// global memory that is read and written with agent-scope accesses only (they bypass the
// per-XCD L2, which is not coherent across XCDs).  There is no counter, no atomic
// read-modify-write and no ordering between stores: the area is made of 8-byte words, each
// holding 32 bits of payload and the 32-bit generation it belongs to (an aligned 8-byte
// store is indivisible), and a reader polls a word until it shows the current generation.
//
// Generation G counts pivot columns across launches (the host passes the first G of a
// launch); buffer G & 1 is used for column G.  Per workgroup and buffer: the candidate
// |a| (1 or 2 words), its row, and the whole panel row of the candidate; per buffer: the
// panel row jj (what the pivot row is swapped with).
template <typename T> struct Words {
  static constexpr int N = sizeof(T) / 4;
};
// Slots are padded to EIGAMD_LU_SLOT_BYTES so that the polls of all workgroups spread over
// the memory channels instead of queueing on the few that a dense array would map to.
#ifndef EIGAMD_LU_SLOT_BYTES
#define EIGAMD_LU_SLOT_BYTES 1024
#endif
template <typename T> struct alignas(EIGAMD_LU_SLOT_BYTES) BaseSlot {
  unsigned long long val[Words<Real<T>>::N];
  unsigned long long row;
  unsigned long long data[kMaxBase * Words<T>::N];
};
template <typename T> struct BaseTop {
  unsigned long long data[kMaxBase * Words<T>::N];
};
constexpr int kMaxFusedParts = 1024;   // workgroups that must be resident together (256 CUs x 4)
constexpr unsigned kSpinLimit = 1u << 20;  // polls of one column before a workgroup gives up
constexpr int kNoRow = 0x7fffffff;

__device__ __forceinline__ unsigned long long coherent_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int coherent_load(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coherent_store(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void put_word(unsigned long long* p, unsigned payload, unsigned G) {
  __hip_atomic_store(p, (unsigned long long)payload | ((unsigned long long)G << 32), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// Value -> words and back.
__device__ __forceinline__ void put_value(unsigned long long* p, double v, unsigned G) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  put_word(p, (unsigned)b, G);
  put_word(p + 1, (unsigned)(b >> 32), G);
}
__device__ __forceinline__ void put_value(unsigned long long* p, float v, unsigned G) {
  put_word(p, __float_as_uint(v), G);
}
// 32-bit word i of a value, in the order put_value stores them.
__device__ __forceinline__ unsigned value_word(double v, int i) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return i == 0 ? (unsigned)b : (unsigned)(b >> 32);
}
__device__ __forceinline__ unsigned value_word(float v, int) { return __float_as_uint(v); }
template <typename R> __device__ __forceinline__ unsigned value_word(Cx<R> v, int i) {
  return i < Words<R>::N ? value_word(v.x, i) : value_word(v.y, i - Words<R>::N);
}
// Reads one value; false while any of its words is not of generation G yet.
__device__ __forceinline__ bool get_value(const unsigned long long* p, double& v, unsigned G) {
  const unsigned long long lo = coherent_load(p), hi = coherent_load(p + 1);
  v = __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
  return (unsigned)(lo >> 32) == G && (unsigned)(hi >> 32) == G;
}
__device__ __forceinline__ bool get_value(const unsigned long long* p, float& v, unsigned G) {
  const unsigned long long w = coherent_load(p);
  v = __uint_as_float((unsigned)w);
  return (unsigned)(w >> 32) == G;
}

template <typename R> __device__ __forceinline__ bool get_value(const unsigned long long* p, Cx<R>& v, unsigned G) {
  const bool re = get_value(p, v.x, G), im = get_value(p + Words<R>::N, v.y, G);
  return re && im;
}

// Largest |a| / smallest row over the workgroup, result in every thread.  One barrier; the
// caller must separate two calls by a barrier of its own (lu_base_fused has one per column).
template <typename T> __device__ Candidate<T> block_best_all(T val, int row) {
  __shared__ T s_val[kRowsPerBlock / 64];
  __shared__ int s_row[kRowsPerBlock / 64];
  wave_best_to_lane63(val, row);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 63) { s_val[wave] = val; s_row[wave] = row; }
  __syncthreads();
  Candidate<T> c = {s_val[0], s_row[0]};
#pragma unroll
  for (int w = 1; w < kRowsPerBlock / 64; ++w)
    if (better(s_val[w], s_row[w], c.val, c.row)) { c.val = s_val[w]; c.row = s_row[w]; }
  return c;
}

// unblocked_lu (PartialPivLU.h:357-412) of the rows x jbw panel at P, jbw <= 16, in one
// launch of ceil(rows / 256) workgroups that must all be resident.  Thread = one row, kept
// in registers from the first load to the final store.  Per column: every workgroup
// publishes its best candidate together with that candidate's row, polls until all
// workgroups have published, then every workgroup picks the same winner and the owners of
// the two rows exchange them.  A workgroup that polls kSpinLimit times without success sets
// *error and leaves; so does every workgroup that then sees the flag: the grid always drains.
// (Four workgroups per compute unit for the real scalars; complex arithmetic needs more registers per row element
// and gets the budget of two.)
template <typename T> constexpr int fused_min_blocks() { return (sizeof(Real<T>) != sizeof(T) ? 2 : 4) * 256 / kRowsPerBlock; }
// WIDE (up to kWideParts workgroups): while wavefront 0 polls the candidates, the other
// three wavefronts read the ROWS of all the slots (and panel row jj) into LDS, so that the winner's row is already
// there when the winner is known and the trip to memory that fetched it (0.76 us of a column's 4.1) leaves the chain.
constexpr int kWideParts = 64;
template <typename T, bool WIDE>
__global__ void __launch_bounds__(kRowsPerBlock, fused_min_blocks<T>())
    lu_base_fused(T* __restrict__ P, int64_t ld, int rows, int jbw, int size, int* __restrict__ transp,
                  int row_offset, int col_global, BaseSlot<T>* __restrict__ slots, BaseTop<T>* __restrict__ tops,
                  unsigned gen0, int* __restrict__ error, int* __restrict__ first_zero, int* __restrict__ nswaps, Ride<T> ride) {
  __shared__ T s_u[kMaxBase];
  __shared__ T s_top[kMaxBase];
  __shared__ unsigned s_pub[kRowsPerBlock / 64][kMaxBase * Words<T>::N];  // per wavefront: the words of the row it is about to publish
  __shared__ int s_win_row, s_flags;   // the winner's row; 1: the pivot is exactly zero, 2: gave up waiting
  __shared__ T s_all[WIDE ? kWideParts + 1 : 1][kMaxBase];   // WIDE: the published row of every slot; [nparts]: panel row jj
  __shared__ int s_gave_up;                                  // WIDE: a wavefront reading the rows gave up waiting
  if (WIDE && threadIdx.x == 0) s_gave_up = 0;               // (read behind the barriers of the first column)
  constexpr int NW = Words<T>::N;          // words of a matrix element
  constexpr int NV = Words<Real<T>>::N;    // words of a candidate's size
  // The last workgroups of the grid take no part in the factorisation: rows of the enclosing panel's inverse and
  // the far part of the block before's update (Ride), in the time the others spend on their sixteen exchanges.
  const int nparts = (int)gridDim.x - ride.inv_groups() - ride.far_groups();
  if ((int)blockIdx.x >= nparts) {
    if (threadIdx.x < 256) run_ride<T>(ride, (int)blockIdx.x - nparts);   // (those workgroups work with 256 threads)
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * kRowsPerBlock + threadIdx.x;
  const bool valid = r < rows;
  T row[kMaxBase];
#pragma unroll
  for (int c = 0; c < kMaxBase; ++c) row[c] = (valid && c < jbw) ? P[r + (int64_t)c * ld] : (T)0;

#ifdef EIGAMD_LU_TIMING
  // Profiling builds: where a pivot column's time goes, summed over the columns of this launch
  // (workgroup 0, thread 0; 10 ns ticks).
  unsigned long long tick[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  unsigned poll_rounds = 0, fetch_rounds = 0;
#define LU_TICK(i) { const unsigned long long now_ = wall_clock64(); tick[i] += now_ - tprev; tprev = now_; }
#else
#define LU_TICK(i)
#endif
  // The loop over the pivot columns is unrolled in full (static_for): every register index is a
  // constant, and column jj updates exactly the columns right of it.
  bool alive = true;  // false once a workgroup has given up waiting
  static_for<kMaxBase>([&](auto jj_c) {
    constexpr int jj = decltype(jj_c)::value;
    if (jj < size && alive) {
#ifdef EIGAMD_LU_TIMING
    tprev = wall_clock64();
#endif
    const unsigned G = gen0 + (unsigned)jj;
    BaseSlot<T>* buf = slots + (size_t)(G & 1u) * kMaxFusedParts;
    BaseTop<T>* top = tops + (G & 1u);
    // candidate of this workgroup for column jj ("no row" when all its rows are above jj)
    const T cur = row[jj];
    const bool cand = valid && r >= jj;
    const Candidate<Real<T>> best = block_best_all<Real<T>>(cand ? absval(cur) : (Real<T>)-1, cand ? r : kNoRow);
    const bool have = best.row != kNoRow;
    LU_TICK(0)
    // The wavefront that holds the candidate's row publishes the slot with one store per lane (two for
    // the few words that a row of complex doubles leaves over): the lane of that row leaves it in LDS (same
    // wavefront: LDS serves its accesses in order, no barrier needed); of the slot's 16 NW + NV + 1 words, taken
    // in the order data, |a|, row index, lane l stores word l (and l + 64).  A workgroup without a candidate
    // publishes "no row" from its first wavefront.
    {
      const int win_local = have ? best.row - (int)blockIdx.x * kRowsPerBlock : 0;
      if ((win_local >> 6) == wave) {
        unsigned* pub = s_pub[wave];
        if (have && lane == (win_local & 63)) {
#pragma unroll
          for (int c = 0; c < kMaxBase; ++c)
#pragma unroll
            for (int w = 0; w < NW; ++w) pub[c * NW + w] = value_word(row[c], w);
        }
        __builtin_amdgcn_wave_barrier();
        unsigned long long* mine = reinterpret_cast<unsigned long long*>(buf + blockIdx.x);
        constexpr int kSlotWords = kMaxBase * NW + NV + 1;
#pragma unroll
        for (int round = 0; round < (kSlotWords + 63) / 64; ++round) {
          const int e = lane + 64 * round;
          unsigned payload = 0;
          int word = -1;  // index into the slot: val[NV], row, data[16 NW]
          if (e < kMaxBase * NW) {
            if (have) {
              payload = pub[e];
              word = NV + 1 + e;
            } else if (WIDE) {   // (the readers of the rows wait for every slot's words)
              word = NV + 1 + e;
            }
          } else if (e < kMaxBase * NW + NV) {
            payload = value_word(best.val, e - kMaxBase * NW);
            word = e - kMaxBase * NW;
          } else if (e == kSlotWords - 1) {
            payload = (unsigned)best.row;
            word = NV;
          }
          if (word >= 0) put_word(mine + word, payload, G);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    // Panel row jj (thread jj of workgroup 0) goes out the same way.
    if (blockIdx.x == 0 && wave == 0) {
      unsigned* pub = s_pub[0];
      if (lane == jj) {
#pragma unroll
        for (int c = 0; c < kMaxBase; ++c)
#pragma unroll
          for (int w = 0; w < NW; ++w) pub[c * NW + w] = value_word(row[c], w);
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < kMaxBase * NW) put_word(&top->data[lane], pub[lane], G);   // (16 NW <= 64)
      __builtin_amdgcn_wave_barrier();
    }
    // Wavefront 0 alone talks to the exchange area: it polls every workgroup's candidate of
    // generation G (lane = slot), picks the winner with shuffles, then fetches the two rows that
    // matter -- panel row jj and, when they are exchanged, the winner's row -- and leaves them in
    // LDS.  No workgroup barrier sits inside a polling loop; the other wavefronts wait at one.
    LU_TICK(1)
    if (threadIdx.x < 64) {
      using R = Real<T>;
      R val = (R)-1;
      int prow = kNoRow;
      unsigned spins = 0;
      bool failed = false;
      // A store takes about 0.4 us to become visible to the other workgroups and a poll that finds
      // nothing costs a whole round trip: the first poll waits a little rather than leave at once.
      __builtin_amdgcn_s_sleep(EIGAMD_LU_POLL_DELAY);
      for (;;) {
        bool ready = true;
        val = (R)-1;
        prow = kNoRow;
#ifdef EIGAMD_LU_TIMING
        ++poll_rounds;
#endif
        for (int i = lane; i < nparts; i += 64) {
          R v;
          const bool okv = get_value(buf[i].val, v, G);
          const unsigned long long w = coherent_load(&buf[i].row);
          if (!okv || (unsigned)(w >> 32) != G) {
            ready = false;
          } else {
            const int rr = (int)(unsigned)w;
            if (better(v, rr, val, prow)) { val = v; prow = rr; }
          }
        }
        if (__all(ready ? 1 : 0)) break;
        ++spins;
        const bool give_up = spins > kSpinLimit || ((spins & 255u) == 0 && coherent_load(error) != 0);
        if (__any(give_up ? 1 : 0)) { failed = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      LU_TICK(2)
      // the same winner in every lane, hence in every workgroup
      wave_best_to_lane63(val, prow);
      val = from_lane63(val);
      prow = from_lane63(prow);
      if (prow == kNoRow) {  // nothing comparable in the column (NaNs): no exchange, pivot = the diagonal entry
        prow = jj;
        val = (R)1;
      }
      const bool zero_w = !(val != (R)0);  // exactly zero
      const bool swap_w = !zero_w && prow != jj;
      if (!WIDE && !failed) {
        const unsigned long long* src_top = top->data;
        const unsigned long long* src_win = buf[swap_w ? prow / kRowsPerBlock : 0].data;
        for (;;) {
          bool ready = true;
          T v = (T)0;
#ifdef EIGAMD_LU_TIMING
          ++fetch_rounds;
#endif
          if (lane < kMaxBase) ready = get_value(&src_top[lane * Words<T>::N], v, G);
          else if (lane < 2 * kMaxBase && swap_w) ready = get_value(&src_win[(lane - kMaxBase) * Words<T>::N], v, G);
          if (__all(ready ? 1 : 0)) {
            if (lane < kMaxBase) {
              s_top[lane] = v;
              if (!swap_w) s_u[lane] = v;
            } else if (lane < 2 * kMaxBase && swap_w) {
              s_u[lane - kMaxBase] = v;
            }
            break;
          }
          ++spins;
          const bool give_up = spins > kSpinLimit || ((spins & 255u) == 0 && coherent_load(error) != 0);
          if (__any(give_up ? 1 : 0)) { failed = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      LU_TICK(3)
      if (lane == 0) {
        s_win_row = prow;
        s_flags = (zero_w ? 1 : 0) | (failed ? 2 : 0);
      }
    } else if (WIDE) {
      // Value c of slot i is item 16 i + c; slot `nparts` stands for panel row jj.  A lane keeps asking for the items
      // it has not seen with generation G yet; nothing in this loop involves another lane.
      const int t = (int)threadIdx.x - 64, total = (nparts + 1) * kMaxBase;
      constexpr int kReaders = kRowsPerBlock - 64;
      constexpr int kItems = ((kWideParts + 1) * kMaxBase + kReaders - 1) / kReaders;
      unsigned pending = 0;
#pragma unroll
      for (int u = 0; u < kItems; ++u)
        if (t + u * kReaders < total) pending |= 1u << u;
      unsigned spins = 0;
      __builtin_amdgcn_s_sleep(EIGAMD_LU_POLL_DELAY);
      while (pending) {
#pragma unroll
        for (int u = 0; u < kItems; ++u) {
          if (pending & (1u << u)) {
            const int e = t + u * kReaders, i = e / kMaxBase, c = e % kMaxBase;
            const unsigned long long* src = (i == nparts ? top->data : buf[i].data) + c * NW;
            T v;
            if (get_value(src, v, G)) {
              s_all[i][c] = v;
              pending &= ~(1u << u);
            }
          }
        }
        if (!pending) break;
        ++spins;
        if (spins > kSpinLimit || ((spins & 255u) == 0 && coherent_load(error) != 0)) {
          atomicOr(&s_gave_up, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    LU_TICK(4)
    if ((s_flags & 2) || (WIDE && s_gave_up)) {
      if (threadIdx.x == 0) coherent_store(error, 1);
      alive = false;
    }
    if (alive) {
    struct { int row; } win;
    win.row = s_win_row;
    const bool zero = (s_flags & 1) != 0;
    const bool swap = !zero && win.row != jj;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      transp[jj] = row_offset + win.row;
      if (zero) atomicCAS(first_zero, -1, col_global + jj);  // keep the FIRST zero pivot (PartialPivLU.h:391-396)
      else if (swap) atomicAdd(nswaps, 1);
    }
    // the pivot row after the exchange, and panel row jj before it
    const T* piv = WIDE ? s_all[swap ? win.row / kRowsPerBlock : nparts] : s_u;
    const T* old = WIDE ? s_all[nparts] : s_top;
    if (swap && valid && (r == win.row || r == jj)) {
      const T* src = r == jj ? piv : old;  // (never both: swap means win.row != jj)
#pragma unroll
      for (int c = 0; c < kMaxBase; ++c) row[c] = src[c];
    }
    if (valid && r > jj) {
      T l = row[jj];
      if (!zero) l = l / piv[jj];  // the reference divides (lu.col(k).tail /= pivot, :389)
      row[jj] = l;
#pragma unroll
      for (int c = jj + 1; c < kMaxBase; ++c) row[c] -= l * piv[c];
    }
    }
    LU_TICK(5)
    // No barrier here: wavefront 0 rewrites s_u / s_top / s_win_row only after the barrier inside
    // the next column's block_best_all, which every wavefront reaches after this update; and the
    // wavefronts rewrite their block_best_all slots only after the barrier above, behind which
    // nobody reads them any more.
    }
  });

  if (valid) {
#pragma unroll
    for (int c = 0; c < kMaxBase; ++c)
      if (c < jbw) P[r + (int64_t)c * ld] = row[c];
  }
#ifdef EIGAMD_LU_TIMING
  if (blockIdx.x == 0 && threadIdx.x == 0 && (col_global & 1023) == 0)
    printf("lu_base_fused parts %d cols %d: reduce %llu publish %llu poll %llu fetch %llu barrier %llu update %llu (x10 ns) poll rounds %u fetch rounds %u\n",
           nparts, size, tick[0], tick[1], tick[2], tick[3], tick[4], tick[5], poll_rounds, fetch_rounds);
#endif
#undef LU_TICK
}

// The same base case for panels of at most single_rows() rows, in ONE workgroup of 512 threads
// with single_rpt() rows per thread: the pivot search and the exchange of the two rows go through
// LDS and two workgroup barriers per column, no global memory and no other workgroup is involved
// (a column costs 1.5 to 2.4 us here, by the number of rows, against 2.7 to 4 us through the exchange
// area).  Same arithmetic as lu_base_fused, operation for operation.
// (Four rows per thread for the real scalars, two for the complex ones.  Measured against 1024 threads
// with two and one -- 2048^2 double 8.1 -> 7.5 ms, complex float 10.9 -> 10.2, the others unchanged -- and
// against 256 threads with eight and four, which is slower throughout: 9.3 ms.)
#ifndef EIGAMD_LU_SINGLE_THREADS
#define EIGAMD_LU_SINGLE_THREADS 512
#endif
constexpr int kSingleThreads = EIGAMD_LU_SINGLE_THREADS;
template <typename T> constexpr int single_rpt() { return (sizeof(Real<T>) != sizeof(T) ? 1024 : 2048) / kSingleThreads; }
template <typename T> constexpr int single_rows() { return kSingleThreads * single_rpt<T>(); }
template <typename T>
__global__ void __launch_bounds__(kSingleThreads)
    lu_base_single(T* __restrict__ P, int64_t ld, int rows, int jbw, int size, int* __restrict__ transp, int row_offset,
                   int col_global, int* __restrict__ first_zero, int* __restrict__ nswaps, Ride<T> ride) {
  if (blockIdx.x > 0) {   // what rides along (Ride): those workgroups work with 256 threads
    if (threadIdx.x < 256) run_ride<T>(ride, (int)blockIdx.x - 1);
    return;
  }
  __shared__ T s_u[kMaxBase];
  __shared__ T s_top[kMaxBase];
  using R = Real<T>;
  constexpr int kSingleRpt = single_rpt<T>();
  __shared__ R s_val[kSingleThreads / 64];
  __shared__ int s_row[kSingleThreads / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // row[q][c]: element c of row tid + q * 1024.  The loop over the pivot columns is unrolled in full,
  // so that every register index is a constant and column jj updates exactly the columns right of it.
  T row[kSingleRpt][kMaxBase];
#pragma unroll
  for (int q = 0; q < kSingleRpt; ++q)
#pragma unroll
    for (int c = 0; c < kMaxBase; ++c) {
      const int r = tid + q * kSingleThreads;
      row[q][c] = (r < rows && c < jbw) ? P[r + (int64_t)c * ld] : (T)0;
    }
  static_for<kMaxBase>([&](auto jj_c) {
    constexpr int jj = decltype(jj_c)::value;
    if (jj < size) {
      R val = (R)-1;
      int prow = kNoRow;
#pragma unroll
      for (int q = 0; q < kSingleRpt; ++q) {
        const int r = tid + q * kSingleThreads;
        if (r < rows && r >= jj) {
          const R v = absval(row[q][jj]);
          if (better(v, r, val, prow)) { val = v; prow = r; }
        }
      }
      wave_best_to_lane63(val, prow);
      if (lane == 63) { s_val[wave] = val; s_row[wave] = prow; }
      __syncthreads();
      val = s_val[0];
      prow = s_row[0];
#pragma unroll
      for (int w = 1; w < kSingleThreads / 64; ++w)
        if (better(s_val[w], s_row[w], val, prow)) { val = s_val[w]; prow = s_row[w]; }
      if (prow == kNoRow) {  // nothing comparable in the column (NaNs): no exchange, pivot = the diagonal entry
        prow = jj;
        val = (R)1;
      }
      const bool zero = !(val != (R)0);  // exactly zero
      const bool swap = !zero && prow != jj;
      // the pivot row (what every row below is updated with) and, for an exchange, panel row jj
#pragma unroll
      for (int q = 0; q < kSingleRpt; ++q) {
        const int r = tid + q * kSingleThreads;
        if (r == jj) {
#pragma unroll
          for (int c = 0; c < kMaxBase; ++c) (swap ? s_top : s_u)[c] = row[q][c];
        } else if (swap && r == prow) {
#pragma unroll
          for (int c = 0; c < kMaxBase; ++c) s_u[c] = row[q][c];
        }
      }
      __syncthreads();
      if (tid == 0) {
        transp[jj] = row_offset + prow;
        if (zero) atomicCAS(first_zero, -1, col_global + jj);  // keep the FIRST zero pivot (PartialPivLU.h:391-396)
        else if (swap) atomicAdd(nswaps, 1);
      }
#pragma unroll
      for (int q = 0; q < kSingleRpt; ++q) {
        const int r = tid + q * kSingleThreads;
        if (swap && (r == jj || r == prow)) {
          const T* src = r == jj ? s_u : s_top;
#pragma unroll
          for (int c = 0; c < kMaxBase; ++c) row[q][c] = src[c];
        }
        if (r < rows && r > jj) {
          T l = row[q][jj];
          if (!zero) l = l / s_u[jj];  // the reference divides (lu.col(k).tail /= pivot, PartialPivLU.h:389)
          row[q][jj] = l;
#pragma unroll
          for (int c = jj + 1; c < kMaxBase; ++c) row[q][c] -= l * s_u[c];
        }
      }
      // s_val / s_row are rewritten before the next column's first barrier and were last read before
      // this column's second one; s_u / s_top are rewritten after the next column's first barrier.
    }
  });
#pragma unroll
  for (int q = 0; q < kSingleRpt; ++q) {
    const int r = tid + q * kSingleThreads;
    if (r < rows) {
#pragma unroll
      for (int c = 0; c < kMaxBase; ++c)
        if (c < jbw) P[r + (int64_t)c * ld] = row[q][c];
    }
  }
}

// After a base case inside a wider panel: the transpositions [k0, k1) (k1 - k0 <= 16) on
// the other columns of the panel -- `nleft` columns starting at A_left and `nright`
// columns starting at A_right -- and, on the right-hand ones, the forward substitution
// with the unit lower triangle L (jb x jb) at rows [k0, k0 + jb).  One thread per column.
template <typename T>
__global__ void lu_rowops(T* __restrict__ A_left, int nleft, T* __restrict__ A_right, int nright, int64_t ld,
                          const int* __restrict__ transp, int k0, int k1, const T* __restrict__ L, int jb) {
  __shared__ T s_l[kMaxBase * kMaxBase];
  __shared__ int s_p[kMaxBase];
  for (int i = threadIdx.x; i < jb * jb; i += blockDim.x)
    s_l[(i % jb) + (i / jb) * kMaxBase] = L[(i % jb) + (int64_t)(i / jb) * ld];
  if ((int)threadIdx.x < k1 - k0) s_p[threadIdx.x] = transp[k0 + threadIdx.x];
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nleft + nright) return;
  const bool right = c >= nleft;
  T* col = right ? A_right + (int64_t)(c - nleft) * ld : A_left + (int64_t)c * ld;
  for (int i = k0; i < k1; ++i) {
    const int p = s_p[i - k0];
    if (p != i) {
      const T t = col[i];
      col[i] = col[p];
      col[p] = t;
    }
  }
  if (!right) return;
  T x[kMaxBase];
#pragma unroll
  for (int i = 0; i < kMaxBase; ++i) x[i] = i < jb ? col[k0 + i] : (T)0;
#pragma unroll
  for (int i = 0; i < kMaxBase; ++i) {
    if (i < jb) {
#pragma unroll
      for (int rr = i + 1; rr < kMaxBase; ++rr)
        if (rr < jb) x[rr] -= s_l[rr + i * kMaxBase] * x[i];
    }
  }
#pragma unroll
  for (int i = 0; i < kMaxBase; ++i)
    if (i < jb) col[k0 + i] = x[i];
}

// info[0] = first zero pivot or -1, info[1] = number of exchanges, info[2] = 1 when a workgroup of the
// one-launch base case gave up waiting: the counters of a factorisation, for callers that do not wait.
__global__ void lu_export_info(const int* __restrict__ header, int* __restrict__ info) {
  info[0] = header[0];
  info[1] = header[1];
  info[2] = header[4];
}

// transp[i] = i: every entry is a valid (identity) transposition before the factorisation
// fills it in, whatever happens to the kernels in between.
__global__ void lu_iota(int* __restrict__ transp, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) transp[i] = i;
}

// Row transpositions i <-> transp[i], i in [k0, k1), applied in order to `ncols`
// columns starting at A; one thread per column.  Row indices are global rows of A.
template <typename T>
__global__ void lu_laswp(T* __restrict__ A, int64_t ld, int ncols, const int* __restrict__ transp, int k0, int k1) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  T* col = A + (int64_t)c * ld;
  for (int i = k0; i < k1; ++i) {
    const int p = transp[i];
    if (p != i) {
      const T t = col[i];
      col[i] = col[p];
      col[p] = t;
    }
  }
}

// C (rows x cols) -= A (rows x jb) * B (jb x cols), jb <= 16: the update of the rest of a
// panel after one base case (PartialPivLU.h:494 at the inner level).  Two flops per 16 bytes
// of C: a streaming kernel -- thread = one row of A in registers, a 64-column slab of B in
// LDS -- in ONE launch, where the GEMM kernel needs an interior launch and edge strips.
template <typename T>
__global__ void __launch_bounds__(256) lu_rank_update(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C,
                                                      int64_t ld, int rows, int cols, int jb) {
  __shared__ T s_b[kMaxBase][kRankCols];
  const int c0 = blockIdx.y * kRankCols;
  const int nc = min(kRankCols, cols - c0);
  for (int idx = threadIdx.x; idx < kMaxBase * kRankCols; idx += 256) {
    const int k = idx / kRankCols, j = idx % kRankCols;
    s_b[k][j] = (k < jb && j < nc) ? B[k + (int64_t)(c0 + j) * ld] : (T)0;
  }
  __syncthreads();
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  T a[kMaxBase];
#pragma unroll
  for (int k = 0; k < kMaxBase; ++k) a[k] = k < jb ? A[r + (int64_t)k * ld] : (T)0;
  T* Cr = C + r + (int64_t)c0 * ld;
  int j0 = 0;
  constexpr int kGroup = EIGAMD_LU_RANK_GROUP;
  for (; j0 + kGroup <= nc; j0 += kGroup) {   // whole groups of columns: independent loads, no tests
    T c[kGroup];
#pragma unroll
    for (int u = 0; u < kGroup; ++u) c[u] = Cr[(int64_t)(j0 + u) * ld];
#pragma unroll
    for (int u = 0; u < kGroup; ++u) {
      T acc = (T)0;
#pragma unroll
      for (int k = 0; k < kMaxBase; ++k) acc += a[k] * s_b[k][j0 + u];
      c[u] -= acc;
    }
#pragma unroll
    for (int u = 0; u < kGroup; ++u) Cr[(int64_t)(j0 + u) * ld] = c[u];
  }
  for (; j0 < nc; ++j0) {
    T acc = (T)0;
#pragma unroll
    for (int k = 0; k < kMaxBase; ++k) acc += a[k] * s_b[k][j0];
    Cr[(int64_t)j0 * ld] -= acc;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) lu_rowops_rank_update(RowopsArgs<T> args) {
  __shared__ RowopsShared<T> sh;
  rowops_rank_update_block<T, EIGAMD_LU_RANK_GROUP, sizeof(Real<T>) == sizeof(T)>(args, (int)blockIdx.x, sh);   // (early loads: real scalars, register budget)
}

// ---- many transpositions at once ----------------------------------------------------
//
// A run of transpositions i <-> transp[i], i in [k0, k1), k1 - k0 <= kPlanSpan, applied in
// order is a permutation of at most 2 (k1 - k0) rows.  lu_swap_plan works it out once as a list
// of moves "row dst receives what row src held" (in no particular order);
// lu_apply_plan then moves whole row segments with independent loads, instead of one
// dependent load-store pair per transposition and column.

__device__ __forceinline__ void swap_plan_block(const int* __restrict__ transp, int k0, int k1, SwapPlan* __restrict__ plan) {
  // Every row that takes part is followed through the run by a thread of its own: the rows
  // [k0, k1) and each row outside them that some transposition names (once, by the first
  // transposition that names it).  The walks are independent of each other, read the run from
  // LDS and keep their row in a register: no barrier inside the run.
  __shared__ int s_p[kPlanSpan];
  __shared__ int s_count;
  const int t = threadIdx.x;
  const int span = k1 - k0;
  if (t < span) s_p[t] = transp[k0 + t];
  if (t == 0) s_count = 0;
  __syncthreads();
  int start[2] = {-1, -1};
  if (t < span) {
    start[0] = k0 + t;
    const int p = s_p[t];
    if (p < k0 || p >= k1) {
      bool first = true;
      for (int j = 0; j < t; ++j) first = first && s_p[j] != p;
      if (first) start[1] = p;
    }
  }
  int at[2] = {start[0], start[1]};
  for (int i = 0; i < span; ++i) {
    const int a = k0 + i, b = s_p[i];
#pragma unroll
    for (int q = 0; q < 2; ++q) at[q] = at[q] == a ? b : (at[q] == b ? a : at[q]);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (start[q] >= 0 && at[q] != start[q]) {  // row at[q] receives what row start[q] held
      const int n = atomicAdd(&s_count, 1);
      plan->dst[n] = at[q];
      plan->src[n] = start[q];
    }
  }
  __syncthreads();
  if (t == 0) plan->count = s_count;
}
__global__ void __launch_bounds__(kPlanSpan) lu_swap_plan(const int* __restrict__ transp, int k0, int k1, SwapPlan* __restrict__ plan) {
  swap_plan_block(transp, k0, k1, plan);
}

// Applies a plan to `ncols` columns starting at A; a workgroup owns kPlanCols columns.  A thread takes one move
// (two when the plan has more than 256) and carries that row's elements of all the workgroup's columns through
// registers: up to 2 * kPlanCols independent loads in flight per thread -- every element is a memory line of its
// own, so the kernel lives on how many lines a compute unit has in flight, and while it holds a compute unit the
// trailing product of the other update stream cannot use it (1024 exchanges of rows 31 744 long: 1.9 ms with 8
// columns per workgroup staged through LDS in a loop of dependent index arithmetic).  Every load of the workgroup
// has completed before any of its stores is issued: a row may be the source of one move and the destination of
// another.
constexpr int kPlanCols = 16;
// (Kernels that only move elements take a complex one as a two-element vector of the hardware: a plain register pair.)
template <typename T> struct RawOf { using type = T; };
template <> struct RawOf<Cx<float>> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct RawOf<Cx<double>> { typedef double type __attribute__((ext_vector_type(2))); };
template <typename T> using Raw = typename RawOf<T>::type;
template <typename T>
__global__ void __launch_bounds__(256) lu_apply_plan(T* __restrict__ A, int64_t ld, int ncols, const SwapPlan* __restrict__ plan) {
  const int n = plan->count;
  const int c0 = blockIdx.x * kPlanCols;
  const int nc = min(kPlanCols, ncols - c0);
  const int t = threadIdx.x;
  T* col0 = A + (int64_t)c0 * ld;
  T v[2][kPlanCols];
  int src[2], dst[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = t + h * 256;
    src[h] = j < n ? plan->src[j] : -1;
    dst[h] = j < n ? plan->dst[j] : -1;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (src[h] >= 0) {
#pragma unroll
      for (int c = 0; c < kPlanCols; ++c)
        if (c < nc) v[h][c] = col0[src[h] + (int64_t)c * ld];
    }
  __builtin_amdgcn_s_waitcnt(0);  // this thread's loads have returned ...
  __syncthreads();                // ... and so have everybody's
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (dst[h] >= 0) {
#pragma unroll
      for (int c = 0; c < kPlanCols; ++c)
        if (c < nc) col0[dst[h] + (int64_t)c * ld] = v[h][c];
    }
}

// B <- L^-1 B for a unit lower-triangular jb x jb block L (jb <= JB) and jb x ncols B:
// one thread per column of B, forward substitution in column (saxpy) order.
template <typename T, int JB>
__global__ void lu_trsm_small(const T* __restrict__ L, int64_t ldl, T* __restrict__ B, int64_t ldb, int jb, int ncols) {
  __shared__ T s_l[JB * JB];
  for (int i = threadIdx.x; i < jb * jb; i += blockDim.x) s_l[(i % jb) + (i / jb) * JB] = L[(i % jb) + (int64_t)(i / jb) * ldl];
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  T x[JB];
  T* col = B + (int64_t)c * ldb;
#pragma unroll
  for (int i = 0; i < JB; ++i) x[i] = i < jb ? col[i] : (T)0;
#pragma unroll
  for (int i = 0; i < JB; ++i) {
    if (i < jb) {
#pragma unroll
      for (int r = i + 1; r < JB; ++r)
        if (r < jb) x[r] -= s_l[r + i * JB] * x[i];
    }
  }
#pragma unroll
  for (int i = 0; i < JB; ++i)
    if (i < jb) col[i] = x[i];
}

// X = L^-1 for the unit lower triangle L of a bs x bs block (bs <= kPlanSpan), written as a
// full bs x bs matrix (zeros above the diagonal) with leading dimension bs.  A workgroup
// owns 32 columns of X, kept in LDS, and walks down the rows 32 at a time: first the part of
// the inner products that only needs rows already done (all 32 x 32 results in parallel,
// independent loads of L), then the 32 x 32 triangle on the diagonal from LDS, one thread
// per column.
constexpr int kInvCols = 32;
template <typename T>
__global__ void __launch_bounds__(256) lu_trtri(const T* __restrict__ L, int64_t ldl, int bs, T* __restrict__ X,
                                                unsigned long long* __restrict__ gmax) {
  __shared__ T s_x[kInvCols][kPlanSpan + 1];   // s_x[c][i]: row i of this workgroup's column c
  __shared__ T s_acc[kInvCols][kInvCols + 1];  // s_acc[c][i - i0]
  __shared__ T s_d[kInvCols][kInvCols + 1];    // s_d[k - i0][i - i0] = L(i, k) of the diagonal block
  const int first = blockIdx.x * kInvCols;      // first column of this workgroup: rows above it are zero
  for (int idx = threadIdx.x; idx < kInvCols * bs; idx += 256) s_x[idx / bs][idx % bs] = (T)0;
  __syncthreads();
  const int il = threadIdx.x % kInvCols;        // row inside the row block
  const int cg = threadIdx.x / kInvCols;        // 8 groups of 4 columns
  for (int i0 = first; i0 < bs; i0 += kInvCols) {
    const int i = i0 + il;
    // (a) acc[c][i] = sum over finished rows k in [first, i0) of L(i, k) * x[c][k]
    T acc[4] = {(T)0, (T)0, (T)0, (T)0};
    if (i < bs) {
      // i0 - first is a multiple of 32: batches of 16 independent loads
      for (int k = first; k < i0; k += 16) {
        T l[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) l[u] = L[i + (int64_t)(k + u) * ldl];
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] += l[u] * s_x[cg * 4 + q][k + u];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) s_acc[cg * 4 + q][il] = acc[q];
    // the diagonal block of L
    for (int idx = threadIdx.x; idx < kInvCols * kInvCols; idx += 256) {
      const int kk = idx / kInvCols, ii = idx % kInvCols;
      s_d[kk][ii] = (i0 + ii < bs && i0 + kk < bs && ii > kk) ? L[(i0 + ii) + (int64_t)(i0 + kk) * ldl] : (T)0;
    }
    __syncthreads();
    // (b) the triangle: thread c walks rows i0 .. i0+31 of its column
    if (threadIdx.x < kInvCols) {
      const int c = threadIdx.x, g = first + c;
      T x[kInvCols];
#pragma unroll
      for (int ii = 0; ii < kInvCols; ++ii) {
        T v = s_acc[c][ii];
#pragma unroll
        for (int kk = 0; kk < kInvCols; ++kk)
          if (kk < ii) v += s_d[kk][ii] * x[kk];
        const int row = i0 + ii;
        x[ii] = (g >= bs || row >= bs || row < g) ? (T)0 : (row == g ? (T)1 : -v);
      }
#pragma unroll
      for (int ii = 0; ii < kInvCols; ++ii)
        if (i0 + ii < bs) s_x[c][i0 + ii] = x[ii];
    }
    __syncthreads();
  }
  double mx = 0.0;
  for (int idx = threadIdx.x; idx < kInvCols * bs; idx += 256) {
    const int c = idx / bs, i = idx % bs;
    if (first + c < bs) {
      X[i + (int64_t)(first + c) * bs] = s_x[c][i];
      mx = abs_max2(mx, (double)absval(s_x[c][i]));
    }
  }
  note_inverse_max(gmax, mx);
}

// The same inverse by block recursion, for bs = 64, 128 or 256 (the usual panel width):
//   inv([A 0; C B]) = [inv(A) 0; -inv(B) C inv(A), inv(B)].
// lu_trtri_diag inverts the 32 x 32 triangles on the diagonal (and writes the zeros above them);
// lu_trtri_merge, once per level h = 32, 64, 128, fills the h x h blocks below pairs of finished
// h x h inverses.  A workgroup owns 32 rows of such a block: it forms T = inv(B)(rows, :) * C in
// LDS (the rows of inv(B) end at the diagonal) and then -T * inv(A), so nothing is updated in
// place and the workgroups of a level are independent.  Every level runs 4 workgroups at
// bs = 256; the chain is 4 short launches against one launch whose first workgroup walks all
// rows (0.29 ms at bs = 256).
template <typename T>
__global__ void __launch_bounds__(64) lu_trtri_diag(const T* __restrict__ L, int64_t ldl, int bs, T* __restrict__ X,
                                                    unsigned long long* __restrict__ gmax) {
  __shared__ T s_d[kInvCols][kInvCols + 1];  // s_d[k][i] = L(i, k) of the diagonal block, i > k
  const int first = blockIdx.x * kInvCols;
  for (int idx = threadIdx.x; idx < kInvCols * kInvCols; idx += 64) {
    const int kk = idx / kInvCols, ii = idx % kInvCols;
    s_d[kk][ii] = ii > kk ? L[(first + ii) + (int64_t)(first + kk) * ldl] : (T)0;
  }
  __syncthreads();
  double mx = 0.0;
  if (threadIdx.x < kInvCols) {
    const int c = threadIdx.x;
    T* col = X + (int64_t)(first + c) * bs;
    for (int i = 0; i < first; ++i) col[i] = (T)0;
    T x[kInvCols];
#pragma unroll
    for (int ii = 0; ii < kInvCols; ++ii) {
      T v = (T)0;
#pragma unroll
      for (int kk = 0; kk < kInvCols; ++kk)
        if (kk < ii) v += s_d[kk][ii] * x[kk];
      x[ii] = ii < c ? (T)0 : (ii == c ? (T)1 : -v);
    }
#pragma unroll
    for (int ii = 0; ii < kInvCols; ++ii) {
      col[first + ii] = x[ii];
      mx = abs_max2(mx, (double)absval(x[ii]));
    }
  }
  note_inverse_max(gmax, mx);  // (outside the branch: the shuffles of the reduction involve every lane)
}

constexpr int kMergeRows = 32, kMergeK = 16, kMergeMaxH = 128;
template <typename T>
__global__ void __launch_bounds__(256) lu_trtri_merge(const T* __restrict__ L, int64_t ldl, int bs, int h, T* __restrict__ X,
                                                      unsigned long long* __restrict__ gmax) {
  __shared__ __attribute__((aligned(16))) T s_t[kMergeRows][kMergeMaxH + 2];   // T = inv(B)(rows, :) * C
  __shared__ __attribute__((aligned(16))) T s_b[kMergeK][kMergeMaxH + 2];      // k-chunk of the right-hand factor
  __shared__ T s_a[kMergeRows][kMergeK + 1];                                   // k-chunk of the rows of inv(B)
  const int per_pair = h / kMergeRows;
  const int pair = blockIdx.x / per_pair, ib = blockIdx.x % per_pair;
  const int base = pair * 2 * h;           // first row / column of A; B starts at base + h
  const int r0 = base + h + ib * kMergeRows;  // first row of this workgroup (in X and in L)
  const int ti = threadIdx.x % 16, tj = threadIdx.x / 16;  // rows ti, ti + 16; columns 2 tj + 32 m + {0, 1}
  const int nm = h / 32;                     // column groups per thread (1, 2 or 4)
  T acc[2][8];

  // phase 0: T(i, j) = sum_k inv(B)(r0 + i, base + h + k) * L(base + h + k, base + j), k < 32 (ib + 1)
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[q][e] = (T)0;
  const int klimit = (ib + 1) * kMergeRows;
  // The k-chunks of both phases go global memory -> registers -> LDS, one chunk ahead: the loads of chunk c + 1 are
  // in flight while chunk c is multiplied (the chain of 16 to 24 chunks was a chain of as many trips to memory).
  constexpr int kBPer = kMergeK * kMergeMaxH / 256, kAPer = kMergeRows * kMergeK / 256;
  T rb[kBPer], ra[kAPer];
  auto fetch0 = [&](int k0) {
#pragma unroll
    for (int u = 0; u < kBPer; ++u) {
      const int idx = threadIdx.x + u * 256;
      const int kk = idx % kMergeK, j = idx / kMergeK;
      rb[u] = idx < kMergeK * h ? L[(base + h + k0 + kk) + (int64_t)(base + j) * ldl] : (T)0;
    }
#pragma unroll
    for (int u = 0; u < kAPer; ++u) {
      const int idx = threadIdx.x + u * 256;
      const int i = idx % kMergeRows, kk = idx / kMergeRows;
      ra[u] = X[(r0 + i) + (int64_t)(base + h + k0 + kk) * bs];
    }
  };
  auto fetch1 = [&](int k0) {
#pragma unroll
    for (int u = 0; u < kBPer; ++u) {
      const int idx = threadIdx.x + u * 256;
      const int kk = idx % kMergeK, j = idx / kMergeK;
      rb[u] = idx < kMergeK * h ? X[(base + k0 + kk) + (int64_t)(base + j) * bs] : (T)0;
    }
  };
  auto put_b = [&]() {
#pragma unroll
    for (int u = 0; u < kBPer; ++u) {
      const int idx = threadIdx.x + u * 256;
      if (idx < kMergeK * h) s_b[idx % kMergeK][idx / kMergeK] = rb[u];
    }
  };
  auto put_a = [&]() {
#pragma unroll
    for (int u = 0; u < kAPer; ++u) {
      const int idx = threadIdx.x + u * 256;
      s_a[idx % kMergeRows][idx / kMergeRows] = ra[u];
    }
  };
  fetch0(0);
  for (int k0 = 0; k0 < klimit; k0 += kMergeK) {
    put_b();
    put_a();
    __syncthreads();
    if (k0 + kMergeK < klimit) fetch0(k0 + kMergeK);
    else fetch1(0);   // the first chunk of phase 1 (inv(A): finished by an earlier launch)
#pragma unroll
    for (int kk = 0; kk < kMergeK; ++kk) {
      const T a0 = s_a[ti][kk], a1 = s_a[ti + 16][kk];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m < nm) {
          const T b0 = s_b[kk][2 * tj + 32 * m], b1 = s_b[kk][2 * tj + 32 * m + 1];
          acc[0][2 * m] += a0 * b0; acc[0][2 * m + 1] += a0 * b1;
          acc[1][2 * m] += a1 * b0; acc[1][2 * m + 1] += a1 * b1;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (m < nm) {
      s_t[ti][2 * tj + 32 * m] = acc[0][2 * m]; s_t[ti][2 * tj + 32 * m + 1] = acc[0][2 * m + 1];
      s_t[ti + 16][2 * tj + 32 * m] = acc[1][2 * m]; s_t[ti + 16][2 * tj + 32 * m + 1] = acc[1][2 * m + 1];
    }
  }
  __syncthreads();

  // phase 1: X(r0 + i, base + j) = - sum_k T(i, k) * inv(A)(base + k, base + j)   (inv(A)(k, j) = 0 for k < j: stored zeros)
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[q][e] = (T)0;
  for (int k0 = 0; k0 < h; k0 += kMergeK) {
    put_b();
    __syncthreads();
    if (k0 + kMergeK < h) fetch1(k0 + kMergeK);
#pragma unroll
    for (int kk = 0; kk < kMergeK; ++kk) {
      const T a0 = s_t[ti][k0 + kk], a1 = s_t[ti + 16][k0 + kk];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m < nm) {
          const T b0 = s_b[kk][2 * tj + 32 * m], b1 = s_b[kk][2 * tj + 32 * m + 1];
          acc[0][2 * m] += a0 * b0; acc[0][2 * m + 1] += a0 * b1;
          acc[1][2 * m] += a1 * b0; acc[1][2 * m + 1] += a1 * b1;
        }
      }
    }
    __syncthreads();
  }
  double mx = 0.0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (m < nm) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = base + 2 * tj + 32 * m + e;
        X[(r0 + ti) + (int64_t)j * bs] = -acc[0][2 * m + e];
        X[(r0 + ti + 16) + (int64_t)j * bs] = -acc[1][2 * m + e];
        mx = abs_max2(abs_max2(mx, (double)absval(acc[0][2 * m + e])), (double)absval(acc[1][2 * m + e]));
      }
    }
  }
  note_inverse_max(gmax, mx);
}

// ---- when the inverse of a unit lower triangle is not small ------------------------------
//
// Partial pivoting bounds |l_ij| by 1, not the entries of inv(L): they are 2 to 4 for the
// triangles a random matrix produces, but can reach 2^(bs-2) (unit lower with -1 everywhere below
// the diagonal), and a product with such an inverse loses what the reference's substitution
// (PartialPivLU.h:492, triangular_solve_matrix) keeps.  The kernels of the inversion note max|inv(L)|
// (note_inverse_max); lu_inverse_finish, right after them: above `limit` (or not finite) it sets *flag and replaces the inverse by the
// identity, and lu_subst_flagged -- launched in front of every product with that inverse, a no-op
// while the flag is clear -- solves by substitution instead.  The launch sequence is the same
// either way, so nothing has to come back to the host.
constexpr double kInverseGrowthLimit = 16.0;
template <typename T>
__global__ void __launch_bounds__(256) lu_inverse_finish(T* __restrict__ X, int bs, double limit,
                                                         unsigned long long* __restrict__ gmax, int* __restrict__ flag) {
  __shared__ int s_subst;
  if (threadIdx.x == 0) {
    const double mx = __longlong_as_double((long long)*gmax);
    *gmax = 0;  // for the next inversion that uses this word
    s_subst = !(mx <= limit);
    *flag = s_subst;
  }
  __syncthreads();
  if (s_subst)
    for (int idx = threadIdx.x; idx < bs * bs; idx += 256) X[idx] = (idx % bs == idx / bs) ? (T)1 : (T)0;
}

// What is left of the inverse behind the panel: its last rows (lu_inverse_rows, kInvGroups workgroups), then the look
// at max|X| (lu_inverse_tail, as lu_inverse_finish).
template <typename T>
__global__ void __launch_bounds__(256) lu_inverse_rows(InvJob<T> job) {
  if ((int)blockIdx.x >= kInvGroups) {   // one more workgroup: the panel's transpositions as one permutation
    swap_plan_block(job.transp, job.k0, job.k1, job.plan);
    return;
  }
  __shared__ InvShared<T> s_inv;
  inverse_rows_job<T, 32>(job, (int)blockIdx.x, s_inv);
}
template <typename T>
__global__ void __launch_bounds__(256) lu_inverse_tail(InvJob<T> job, double limit, int* __restrict__ flag) {
  __shared__ int s_subst;
  if (threadIdx.x == 0) {
    const unsigned long long b = __hip_atomic_load(job.gmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double mx = __longlong_as_double((long long)b);
    __hip_atomic_store(job.gmax, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next inversion that uses this word
    s_subst = !(mx <= limit);
    *flag = s_subst;
  }
  __syncthreads();
  if (s_subst)
    for (int idx = threadIdx.x; idx < job.bs * job.bs; idx += 256) job.X[idx] = (idx % job.bs == idx / job.bs) ? (T)1 : (T)0;
}

// B (bs x ncols) <- L^-1 B by substitution, L unit lower (bs <= kPlanSpan), one thread per column,
// the unknowns 32 at a time: their 32 x 32 triangle from LDS, then one read-modify-write per row below.
template <typename T>
__global__ void __launch_bounds__(256) lu_subst_flagged(const int* __restrict__ flag, const T* __restrict__ L, int64_t ldl, int bs,
                                                        T* __restrict__ B, int64_t ldb, int ncols) {
  if (*flag == 0) return;
  constexpr int CH = 32;
  __shared__ T s_t[CH][kPlanSpan + 1];  // s_t[kk][i - i0] = L(i, i0 + kk), i > i0 + kk
  const int col = blockIdx.x * 256 + threadIdx.x;
  T* b = B + (int64_t)col * ldb;
  for (int i0 = 0; i0 < bs; i0 += CH) {
    const int w = min(CH, bs - i0);
    for (int idx = threadIdx.x; idx < CH * (bs - i0); idx += 256) {
      const int kk = idx % CH, ii = idx / CH;
      s_t[kk][ii] = (kk < w && ii > kk) ? L[(i0 + ii) + (int64_t)(i0 + kk) * ldl] : (T)0;
    }
    __syncthreads();
    if (col < ncols) {
      T x[CH];
#pragma unroll
      for (int ii = 0; ii < CH; ++ii) x[ii] = ii < w ? b[i0 + ii] : (T)0;
#pragma unroll
      for (int ii = 0; ii < CH; ++ii) {
        T v = x[ii];
#pragma unroll
        for (int kk = 0; kk < CH; ++kk)
          if (kk < ii) v -= s_t[kk][ii] * x[kk];
        x[ii] = v;
      }
#pragma unroll
      for (int ii = 0; ii < CH; ++ii)
        if (ii < w) b[i0 + ii] = x[ii];
      for (int i = i0 + CH; i < bs; ++i) {
        T acc = (T)0;
#pragma unroll
        for (int kk = 0; kk < CH; ++kk) acc += s_t[kk][i - i0] * x[kk];
        b[i] -= acc;
      }
    }
    __syncthreads();
  }
}

template <typename T> struct KindOf;
template <> struct KindOf<float> { static constexpr int value = F32; };
template <> struct KindOf<double> { static constexpr int value = F64; };
template <> struct KindOf<Cx<float>> { static constexpr int value = C32; };
template <> struct KindOf<Cx<double>> { static constexpr int value = C64; };
inline hipError_t launch_product(const GemmCall& c, hipStream_t s) {
  switch (c.kind) {
    case F32: return launch_gemm_f32(c, s);
    case F64: return launch_gemm_f64(c, s);
    case C32: return launch_gemm_c32(c, s);
    default:  return launch_gemm_c64(c, s);
  }
}

template <typename T> hipError_t gemm_minus3(int64_t m, int64_t n, int64_t k, const T* A, int64_t lda, const T* B,
                                             int64_t ldb, T* C, int64_t ldc, hipStream_t s);
// EIGAMD_LU_ATOMIC=0: the trailing updates read, add and store C instead of adding at the memory side (experiments).
inline bool atomic_updates_enabled() {
  static const bool on = [] { const char* e = std::getenv("EIGAMD_LU_ATOMIC"); return !(e && std::atoi(e) == 0); }();
  return on;
}

template <typename T> hipError_t gemm_minus(int64_t m, int64_t n, int64_t k, const T* A, const T* B, T* C, int64_t ld,
                                            hipStream_t s) {
  return gemm_minus3<T>(m, n, k, A, ld, B, ld, C, ld, s);
}

template <typename T> hipError_t gemm_minus3(int64_t m, int64_t n, int64_t k, const T* A, int64_t lda, const T* B,
                                             int64_t ldb, T* C, int64_t ldc, hipStream_t s) {
  if (m <= 0 || n <= 0 || k <= 0) return hipSuccess;
  GemmCall c;
  c.kind = KindOf<T>::value;
  c.opa = OP_N; c.opb = OP_N;
  c.m = m; c.n = n; c.k = k;
  c.serial_edges = true;   // the look-ahead owns the CU split between its streams
  // (a k = 256 product is bound by the rate of the atomic adds: 53 against 55 TFLOP/s; real scalars only)
  c.atomic_c = atomic_updates_enabled() && k >= 512 && sizeof(Real<T>) == sizeof(T);
  c.alpha[0] = -1.0; c.alpha[1] = 0.0;
  c.beta[0] = 1.0; c.beta[1] = 0.0;
  c.A = A; c.lda = lda;
  c.B = B; c.ldb = ldb;
  c.C = C; c.ldc = ldc;
  return launch_product(c, s);
}

// C = A * B + beta * C (alpha = 1; beta = 0 or 1).
template <typename T> hipError_t gemm_assign(int64_t m, int64_t n, int64_t k, const T* A, int64_t lda, const T* B, int64_t ldb,
                                             T* C, int64_t ldc, hipStream_t s, double beta = 0.0) {
  if (m <= 0 || n <= 0) return hipSuccess;
  GemmCall c;
  c.kind = KindOf<T>::value;
  c.opa = OP_N; c.opb = OP_N;
  c.m = m; c.n = n; c.k = k;
  c.serial_edges = true;
  c.alpha[0] = 1.0; c.alpha[1] = 0.0;
  c.beta[0] = beta; c.beta[1] = 0.0;
  c.A = A; c.lda = lda;
  c.B = B; c.ldb = ldb;
  c.C = C; c.ldc = ldc;
  return launch_product(c, s);
}

int block_size_rule(int64_t size, int64_t max_block) {  // PartialPivLU.h:443-448
  int64_t b = size / 8;
  b = (b / 16) * 16;
  return (int)std::min<int64_t>(std::max<int64_t>(b, 8), max_block);
}

#define LU_TRY(expr)                         \
  do {                                       \
    hipError_t e_ = (expr);                  \
    if (e_ != hipSuccess) return e_;         \
  } while (0)

// Which base case a panel may take, read from the environment at every factorisation (so that a
// test can walk all three): EIGAMD_LU_SINGLE=0 keeps short panels off the one-workgroup kernel,
// EIGAMD_LU_FUSED=0 keeps panels off the one-launch kernel (the chain of per-column kernels runs).
struct BaseCaseRule {
  bool single = true, fused = true;
};
inline BaseCaseRule read_base_case_rule() {
  BaseCaseRule r;
  if (const char* e = std::getenv("EIGAMD_LU_SINGLE")) r.single = std::atoi(e) != 0;
  if (const char* e = std::getenv("EIGAMD_LU_FUSED")) r.fused = std::atoi(e) != 0;
  return r;
}

// Where the inverse of a panel's unit lower triangle is to be built while the panel is factorised (InvJob).
template <typename T> struct InverseTarget {
  T* X = nullptr;
  int* flag = nullptr;
  unsigned long long* gmax = nullptr;
  T* XT = nullptr;   // scratch for the transpose (InvJob)
  SwapPlan* plan = nullptr;   // when set: the launch behind the panel also turns the panel's transpositions into this plan
};

template <typename T> struct Workspace {
  Candidate<Real<T>>* partials;
  Side<T>* side;
  int* first_zero;
  int* nswaps;
  BaseSlot<T>* slots;   // 2 x ceil(rows / 256) slots of the fused base case
  BaseTop<T>* tops;     // 2
  unsigned* generation; // host-side running count of pivot columns handed to lu_base_fused
  int* error;
  int fused_capacity;   // workgroups of lu_base_fused that are certainly resident together
  int wide_poll_min = 2;  // lu_base_fused reads all rows beside the poll from this many workgroups on (EIGAMD_LU_WIDE_POLL=0: never)
  BaseCaseRule rule;
  int64_t wide_fixed, wide_t4, wide_t2;  // outer-step width rule of the look-ahead (outer_block)
  int64_t slice_cols, slice_min;  // the panel stream updates the last slice_cols columns itself while >= slice_min remain (0: off)
  int64_t slice_from;             // ... in factorisations of at least this many columns beside the slice
  bool ramp;            // the first outer steps of the look-ahead are narrower than the rule's width
  int64_t split_min;    // the bulk update is dealt over two streams while at least this many columns remain (0: never)
  // Scratch of lu_swap_plan / lu_apply_plan and inverses of unit lower triangles (kPlanSpan^2
  // each).  Sets [2 * kMaxSub): the look-ahead, alternating per outer step, one entry per
  // kPlanSpan-wide piece of the step's block; the last set: blocked() itself (one stream).
  SwapPlan* plan[2 * kMaxSub + 1];
  T* linv[2 * kMaxSub + 1];
  T* linvT[2 * kMaxSub + 1];        // their transposes while they are built (InvJob)
  T* solve_tmp;                     // (kMaxSub kPlanSpan)^2 elements: the solved rows of the next step's columns (see blocked_lookahead)
  int* linv_flag[2 * kMaxSub + 1];  // set by lu_inverse_finish: solve by substitution, linv holds the identity
  unsigned long long* linv_max[2 * kMaxSub + 1];  // max|inv(L)| as the inversion kernels noted it (zero between uses)
  double inverse_limit;
  int* rank_count;      // per column block of lu_rowops_rank_update (zero between launches), or null: two launches
  int64_t rank_fused_max;  // ... used for panels of at most this many rows below the block (taller: two launches)
  int64_t far_rides_max;   // the far part of a block's update rides in the next base-case launch up to this many rows below the block
};
constexpr int kOwnScratch = 2 * kMaxSub;

// unblocked_lu (PartialPivLU.h:357-412) on the rows x jbw sub-panel at P.
template <typename T>
hipError_t unblocked(T* P, int64_t ld, int64_t rows, int jbw, int* transp, int row_offset, int col_offset,
                     const Workspace<T>& w, hipStream_t s, const Ride<T>& ride = Ride<T>()) {
  const int size = (int)std::min<int64_t>(rows, jbw);
  int nparts = (int)((rows + kRowsPerBlock - 1) / kRowsPerBlock);
  const int njob = ride.inv_groups() + ride.far_groups();   // extra workgroups at the end of the grid (they wait for nobody)
  if (rows <= single_rows<T>() && w.rule.single) {
    hipLaunchKernelGGL(lu_base_single<T>, dim3(1 + njob), dim3(kSingleThreads), 0, s, P, ld, (int)rows, jbw, size, transp,
                       row_offset, col_offset, w.first_zero, w.nswaps, ride);
    return hipGetLastError();
  }
  if (nparts <= w.fused_capacity && w.rule.fused) {   // (the base case's own workgroups are dispatched first)
    const unsigned gen0 = *w.generation;
    *w.generation += (unsigned)size;
    if (nparts <= kWideParts && nparts >= w.wide_poll_min)
      hipLaunchKernelGGL((lu_base_fused<T, true>), dim3(nparts + njob), dim3(kRowsPerBlock), 0, s, P, ld, (int)rows, jbw, size, transp,
                         row_offset, col_offset, w.slots, w.tops, gen0, w.error, w.first_zero, w.nswaps, ride);
    else
      hipLaunchKernelGGL((lu_base_fused<T, false>), dim3(nparts + njob), dim3(kRowsPerBlock), 0, s, P, ld, (int)rows, jbw, size, transp,
                         row_offset, col_offset, w.slots, w.tops, gen0, w.error, w.first_zero, w.nswaps, ride);
    return hipGetLastError();
  }
  if (ride.far_groups()) hipLaunchKernelGGL(lu_rowops_rank_update<T>, dim3((unsigned)ride.far_groups()), dim3(256), 0, s, ride.far);
  if (ride.inv_groups()) hipLaunchKernelGGL(lu_inverse_rows<T>, dim3(kInvGroups), dim3(256), 0, s, ride.inv);
  hipLaunchKernelGGL(lu_colmax<T>, dim3(nparts), dim3(kRowsPerBlock), 0, s, P, ld, 0, (int)rows, 0, w.partials);
  for (int jj = 0; jj < size; ++jj) {
    hipLaunchKernelGGL(lu_pivot<T>, dim3(1), dim3(64), 0, s, P, ld, jbw, jj, w.partials, nparts, transp + jj, row_offset,
                       col_offset + jj, w.side, w.first_zero, w.nswaps);
    const int64_t below = rows - jj - 1;
    if (below > 0) {  // (:398) k < rows-1
      nparts = (int)((below + kRowsPerBlock - 1) / kRowsPerBlock);
      hipLaunchKernelGGL(lu_update<T>, dim3(nparts), dim3(kRowsPerBlock), 0, s, P, ld, (int)rows, jbw, jj, w.side,
                         w.partials);
    }
  }
  return hipGetLastError();
}

hipError_t make_plan(const int* transp, int k0, int k1, SwapPlan* plan, hipStream_t s) {
  hipLaunchKernelGGL(lu_swap_plan, dim3(1), dim3(kPlanSpan), 0, s, transp, k0, k1, plan);
  return hipGetLastError();
}
template <typename T> hipError_t apply_plan(T* A, int64_t ld, int64_t ncols, const SwapPlan* plan, hipStream_t s) {
  if (ncols <= 0) return hipSuccess;
  hipLaunchKernelGGL(lu_apply_plan<Raw<T>>, dim3((unsigned)((ncols + kPlanCols - 1) / kPlanCols)), dim3(256), 0, s,
                     reinterpret_cast<Raw<T>*>(A), ld, (int)ncols, plan);
  return hipGetLastError();
}

// Short runs: one thread per column applies them one after the other.  Long runs: by plan,
// kPlanSpan transpositions at a time (`plan` is device scratch of the calling stream).
template <typename T>
hipError_t laswp(T* A, int64_t ld, int64_t ncols, const int* transp, int k0, int k1, SwapPlan* plan, hipStream_t s) {
  if (ncols <= 0 || k1 <= k0) return hipSuccess;
  if (k1 - k0 <= 32 || plan == nullptr) {
    hipLaunchKernelGGL(lu_laswp<Raw<T>>, dim3((unsigned)((ncols + 127) / 128)), dim3(128), 0, s, reinterpret_cast<Raw<T>*>(A), ld,
                       (int)ncols, transp, k0, k1);
    return hipGetLastError();
  }
  for (int k = k0; k < k1; k += kPlanSpan) {
    LU_TRY(make_plan(transp, k, std::min(k + kPlanSpan, k1), plan, s));
    LU_TRY(apply_plan<T>(A, ld, ncols, plan, s));
  }
  return hipSuccess;
}

// B (bs x ncols) <- L^-1 B with L the unit lower triangle of the bs x bs block at L:
// 32-row steps of forward substitution, the rows below updated by the GEMM kernel.
template <typename T>
hipError_t trsm_unit_lower2(const T* L, int64_t ldl, T* B, int64_t ldb, int64_t bs, int64_t ncols, hipStream_t s) {
  if (ncols <= 0) return hipSuccess;
  constexpr int JB = 32;
  for (int64_t j = 0; j < bs; j += JB) {
    const int jb = (int)std::min<int64_t>(JB, bs - j);
    hipLaunchKernelGGL((lu_trsm_small<T, JB>), dim3((unsigned)((ncols + 63) / 64)), dim3(64), 0, s, L + j + j * ldl, ldl,
                       B + j, ldb, jb, (int)ncols);
    LU_TRY(gemm_minus3<T>(bs - j - jb, ncols, jb, L + (j + jb) + j * ldl, ldl, B + j, ldb, B + j + jb, ldb, s));
  }
  return hipGetLastError();
}
template <typename T>
hipError_t trsm_unit_lower(const T* L, T* B, int64_t ld, int64_t bs, int64_t ncols, hipStream_t s) {
  return trsm_unit_lower2<T>(L, ld, B, ld, bs, ncols, s);
}

// The same solve as ONE product with the explicit inverse of the triangle (|l_ij| <= 1 under
// partial pivoting): B <- X * B, in place.  In place is safe because the whole row extent
// fits one tile row of the GEMM (bs <= BM): a workgroup reads only the columns of B that it
// overwrites, and it reads all of them before its epilogue stores.
template <typename T> constexpr int gemm_tile_rows() { return Geometry<KindOf<T>::value>::BM; }
inline bool recursive_inverse_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("EIGAMD_LU_TRTRI_RECURSIVE");
    return !(e && std::atoi(e) == 0);
  }();
  return on;
}
// (lu_trtri keeps 32 whole columns of X in LDS, which complex doubles do not fit: they take the recursion, and
// inverse_applies() admits no other width for them.)
template <typename T> constexpr bool one_launch_inverse_fits() { return sizeof(T) <= 8; }
template <typename T>
hipError_t make_inverse(const T* L, int64_t ldl, int64_t bs, T* X, int* flag, unsigned long long* gmax, double limit,
                        hipStream_t s) {
  if ((bs == 64 || bs == 128 || bs == 256) && (recursive_inverse_enabled() || !one_launch_inverse_fits<T>())) {
    hipLaunchKernelGGL(lu_trtri_diag<T>, dim3((unsigned)(bs / kInvCols)), dim3(64), 0, s, L, ldl, (int)bs, X, gmax);
    for (int h = kInvCols; h < bs; h *= 2)
      hipLaunchKernelGGL(lu_trtri_merge<T>, dim3((unsigned)(bs / 64)), dim3(256), 0, s, L, ldl, (int)bs, h, X, gmax);
  } else if constexpr (one_launch_inverse_fits<T>()) {
    hipLaunchKernelGGL(lu_trtri<T>, dim3((unsigned)((bs + kInvCols - 1) / kInvCols)), dim3(256), 0, s, L, ldl, (int)bs, X, gmax);
  } else {
    return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(lu_inverse_finish<T>, dim3(1), dim3(256), 0, s, X, (int)bs, limit, gmax, flag);
  return hipGetLastError();
}
inline bool far_rides_enabled() {   // EIGAMD_LU_FAR_RIDES=0: a block's whole update in the launch between two base cases
  static const bool on = [] {
    const char* e = std::getenv("EIGAMD_LU_FAR_RIDES");
    return !(e && std::atoi(e) == 0);
  }();
  return on;
}
inline bool solve_out_of_place_enabled() {   // EIGAMD_LU_SOLVE_TMP=0: the next step's columns are solved in place
  static const bool on = [] {
    const char* e = std::getenv("EIGAMD_LU_SOLVE_TMP");
    return !(e && std::atoi(e) == 0);
  }();
  return on;
}
inline bool incremental_inverse_enabled() {   // EIGAMD_LU_INVERSE_ON_THE_WAY=0: the chain of launches behind the panel (make_inverse)
  static const bool on = [] {
    const char* e = std::getenv("EIGAMD_LU_INVERSE_ON_THE_WAY");
    return !(e && std::atoi(e) == 0);
  }();
  return on;
}
// (A block taller than one tile row of the kind's product -- complex: 128 -- is multiplied in two halves, see
// apply_inverse; hence up to two tile rows.)
template <typename T> bool inverse_applies(int64_t bs) {
  if (!one_launch_inverse_fits<T>() && bs != 64 && bs != 128 && bs != 256) return false;
  return bs > 32 && bs <= kPlanSpan && bs <= 2 * gemm_tile_rows<T>();
}
// With `out` the product goes there instead (leading dimension ldo): a product that does not overwrite its operand may
// run on any tile, and a block of few columns then spreads over many compute units instead of two large tiles.
template <typename T>
hipError_t apply_inverse(const T* X, const int* flag, const T* L, int64_t ldl, int64_t bs, T* B, int64_t ldb, int64_t ncols,
                         hipStream_t s, T* out = nullptr, int64_t ldo = 0) {
  if (ncols <= 0) return hipSuccess;
  hipLaunchKernelGGL(lu_subst_flagged<T>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, flag, L, ldl, (int)bs, B, ldb,
                     (int)ncols);
  LU_TRY(hipGetLastError());
  if (out) return gemm_assign<T>(bs, ncols, bs, X, bs, B, ldb, out, ldo, s);
  const int64_t h = gemm_tile_rows<T>();
  if (bs <= h) return gemm_assign<T>(bs, ncols, bs, X, bs, B, ldb, B, ldb, s);
  // Two tile rows, in place: with X = [X11 0; X21 X22] and B = [B1; B2], first B2 <- X22 B2 (one tile row, in place),
  // then B2 += X21 B1 (B1 still holds the operand), then B1 <- X11 B1.
  LU_TRY(gemm_assign<T>(bs - h, ncols, bs - h, X + h + h * bs, bs, B + h, ldb, B + h, ldb, s));
  LU_TRY(gemm_assign<T>(bs - h, ncols, h, X + h, bs, B, ldb, B + h, ldb, s, 1.0));
  return gemm_assign<T>(h, ncols, h, X, bs, B, ldb, B, ldb, s);
}

// blocked_lu (PartialPivLU.h:429-498).  A points at the (row0, row0) corner of the
// enclosing matrix `base`; transposition entries are written as GLOBAL row indices of
// `base`, so the caller-level "+= k" of the reference (:481) is already applied.
// `inv`: at the level of a panel (blocks of 16) the inverse of the panel's triangle is built into inv[0] on the way
// (complete, with its flag, when the call returns); at a level above, inv[i] is the target of the i-th panel of
// this level (null: the scratch of blocked() itself), which the solve behind that panel then uses.
template <typename T>
hipError_t blocked(T* base, int64_t ld, int64_t row0, int64_t col0, int64_t rows, int64_t cols, int* transp,
                   int64_t max_block, const Workspace<T>& w, hipStream_t s, int64_t force_block = 0,
                   const InverseTarget<T>* inv = nullptr, const Ride<T>& ride = Ride<T>()) {
  T* A = base + row0 + col0 * ld;
  const int64_t size = std::min(rows, cols);
  if (size <= kMaxBase) {
    // unblocked_lu works on whole rows.  For a block wider than it is tall, factorise the
    // leading rows x rows part and bring the columns to its right up to date with the same
    // operations in the same order: the recorded swaps, then forward substitution.
    const int jbw = (int)std::min<int64_t>(cols, rows);
    LU_TRY(unblocked<T>(A, ld, rows, jbw, transp + row0, (int)row0, (int)col0, w, s, ride));
    if (cols > jbw) {
      LU_TRY(laswp<T>(base + (col0 + jbw) * ld, ld, cols - jbw, transp, (int)row0, (int)(row0 + jbw), w.plan[kOwnScratch], s));
      hipLaunchKernelGGL((lu_trsm_small<T, kMaxBase>), dim3((unsigned)((cols - jbw + 63) / 64)), dim3(64), 0, s, A, ld,
                         A + (int64_t)jbw * ld, ld, jbw, (int)(cols - jbw));
    }
    return hipGetLastError();
  }
  const int64_t block = force_block > 0 ? force_block : block_size_rule(size, max_block);
  // (panel level) rows of the inverse that are there; the job of a launch: the rows whose part of L is final by then
  const bool build_inverse = inv != nullptr && block <= kMaxBase && inverse_applies<T>(size);
  int64_t x_done = 0;
  auto inverse_job = [&](int64_t upto) {
    InvJob<T> job;
    if (build_inverse && upto > x_done) {
      job.L = A; job.ldl = ld; job.X = inv[0].X; job.XT = inv[0].XT; job.gmax = inv[0].gmax;
      job.bs = (int)size; job.i0 = (int)x_done; job.i1 = (int)upto;
      x_done = upto;
    }
    return job;
  };
  const InverseTarget<T> own = {w.linv[kOwnScratch], w.linv_flag[kOwnScratch], w.linv_max[kOwnScratch], w.linvT[kOwnScratch]};
  // (panel level) the far part of the block before's update: it rides in the launch of this block's base case
  RowopsArgs<T> pending_far;
  for (int64_t k = 0; k < size; k += block) {
    const int64_t bs = std::min(size - k, block);
    const int64_t trows = rows - k - bs, tsize = size - k - bs;
    const InverseTarget<T>& target = (block > kMaxBase && inv != nullptr) ? inv[k / block] : own;
    const bool inverted_on_the_way = block > kMaxBase && trows > 0 && inverse_applies<T>(bs) && incremental_inverse_enabled();
    // the panel [A11; A21], recursively with blocks of 16 (:472-473); at panel level the launch of this block's base
    // case carries the rows of the inverse that belong to the blocks before it (their part of L no longer moves)
    Ride<T> carried;
    carried.inv = inverse_job(k);
    carried.far = pending_far;
    pending_far = RowopsArgs<T>();
    LU_TRY(blocked<T>(base, ld, row0 + k, col0 + k, trows + bs, bs, transp, 16, w, s, 0, inverted_on_the_way ? &target : nullptr, carried));
    if (bs <= kMaxBase) {
      // Narrow blocks (inside a panel): the transpositions on both sides (A_0 :479-483, A_2
      // :488-489) and the triangular solve (:492) in one launch, then the rank-bs update (:494).
      const int64_t nright = trows ? tsize : 0;
      // Row operations and rank update in one launch: one arrival counter per 32-column block, kRankCounters of
      // them in the header.  A panel of the look-ahead has at most kPlanSpan columns right of a block; a caller
      // that hands a wide matrix to blocked() with blocks of 16 (max_block_size <= 16) may have more, and takes
      // the two launches.
      const int ncb = (int)((tsize + kRankCols - 1) / kRankCols);
      const bool far_rides = far_rides_enabled() && ncb <= kRankCounters && trows <= w.far_rides_max;
      if (w.rank_count && trows > 0 && (trows <= w.rank_fused_max || far_rides) && tsize > 0 && ncb <= kRankCounters) {
        RowopsArgs<T> ra;
        ra.base = base; ra.ld = ld; ra.transp = transp;
        ra.k0 = (int)(row0 + k); ra.jb = (int)bs; ra.cL = (int)(col0 + k); ra.cR = (int)(col0 + k + bs); ra.nright = (int)tsize;
        ra.cLeft0 = (int)col0; ra.nleft = (int)k; ra.trows = (int)trows;
        ra.nrb = (int)((trows + 255) / 256); ra.ncb = ncb; ra.counters = w.rank_count;
        if (far_rides) {
          // Only the next block's columns stand between this block and the next base case; the columns beyond them
          // are updated by extra workgroups of that base case's launch (same operations per element, same order:
          // that launch comes before the next block's own row operations).
          const int near = (int)std::min<int64_t>(block, tsize);
          pending_far = ra;
          pending_far.cR = ra.cR + near; pending_far.nright = ra.nright - near; pending_far.nleft = 0;
          pending_far.ncb = (pending_far.nright + kRankCols - 1) / kRankCols;
          ra.nright = near; ra.ncb = 1;
        }
        hipLaunchKernelGGL(lu_rowops_rank_update<T>, dim3((unsigned)ra.workgroups()), dim3(256), 0, s, ra);
        LU_TRY(hipGetLastError());
        continue;
      }
      if (k + nright > 0) {
        hipLaunchKernelGGL(lu_rowops<T>, dim3((unsigned)((k + nright + 63) / 64)), dim3(64), 0, s, base + col0 * ld, (int)k,
                           base + (col0 + k + bs) * ld, (int)nright, ld, transp, (int)(row0 + k), (int)(row0 + k + bs),
                           A + k + k * ld, (int)bs);
        LU_TRY(hipGetLastError());
      }
      if (trows && tsize > 0) {
        hipLaunchKernelGGL(lu_rank_update<T>,
                           dim3((unsigned)((trows + 255) / 256), (unsigned)((tsize + kRankCols - 1) / kRankCols)),
                           dim3(256), 0, s, A + (k + bs) + k * ld, A + k + (k + bs) * ld, A + (k + bs) + (k + bs) * ld, ld,
                           (int)trows, (int)tsize, (int)bs);
        LU_TRY(hipGetLastError());
      }
      if (build_inverse && k + bs >= size) {   // the panel's last block: what is left of the inverse, and the look at max|X|
        InvJob<T> job = inverse_job(size);
        if (inv[0].plan) { job.plan = inv[0].plan; job.transp = transp; job.k0 = (int)row0; job.k1 = (int)(row0 + size); }
        if (job.i1 > job.i0) hipLaunchKernelGGL(lu_inverse_rows<T>, dim3(kInvGroups + (job.plan ? 1 : 0)), dim3(256), 0, s, job);
        else if (job.plan) LU_TRY(make_plan(transp, job.k0, job.k1, job.plan, s));
        job.X = inv[0].X; job.gmax = inv[0].gmax; job.bs = (int)size;
        hipLaunchKernelGGL(lu_inverse_tail<T>, dim3(1), dim3(256), 0, s, job, w.inverse_limit, inv[0].flag);
        LU_TRY(hipGetLastError());
      }
      continue;
    }
    // transpositions of the panel on the columns to its left inside this level (A_0, :479-483)
    LU_TRY(laswp<T>(base + col0 * ld, ld, k, transp, (int)(row0 + k), (int)(row0 + k + bs), w.plan[kOwnScratch], s));
    if (trows) {
      // ... and on the columns to its right (A_2, :488-489)
      LU_TRY(laswp<T>(base + (col0 + k + bs) * ld, ld, tsize, transp, (int)(row0 + k), (int)(row0 + k + bs), w.plan[kOwnScratch], s));
      T* A11 = A + k + k * ld;
      T* A12 = A + k + (k + bs) * ld;
      T* A21 = A + (k + bs) + k * ld;
      T* A22 = A + (k + bs) + (k + bs) * ld;
      if (inverse_applies<T>(bs)) {                                    // (:492)
        if (!inverted_on_the_way) LU_TRY(make_inverse<T>(A11, ld, bs, target.X, target.flag, target.gmax, w.inverse_limit, s));
        LU_TRY(apply_inverse<T>(target.X, target.flag, A11, ld, bs, A12, ld, tsize, s));
      } else {
        LU_TRY(trsm_unit_lower<T>(A11, A12, ld, bs, tsize, s));
      }
      LU_TRY(gemm_minus<T>(trows, tsize, bs, A21, A12, A22, ld, s));  // (:494) the GEMM hot path
    }
  }
  if (pending_far.nright > 0) {   // (a far part that found no base case to ride in)
    hipLaunchKernelGGL(lu_rowops_rank_update<T>, dim3((unsigned)pending_far.workgroups()), dim3(256), 0, s, pending_far);
    LU_TRY(hipGetLastError());
  }
  return hipSuccess;
}

// Top level of blocked_lu with look-ahead: the panel of step k+1 is factorised on the
// `panel` stream while the bulk of the trailing update of step k still runs on the
// `update` stream.  Per step, on the update stream and in this order: the columns of
// the NEXT panel are brought up to date first (transpositions, triangular solve, GEMM)
// and an event is recorded; then the transpositions of the columns to the left, and
// transpositions / solve / GEMM of the remaining trailing columns.  The panel stream
// waits for that event only.  The arithmetic and its order per element are those of
// the sequential algorithm; only independent pieces overlap:
//  * the panel of step k+1 touches columns [k+bs, k+2bs), rows >= k+bs: written by the
//    "next panel" GEMM before the event, never touched by the remaining GEMM of step k;
//  * the remaining GEMM of step k reads A21 of panel k (columns [k, k+bs)), which the
//    transpositions of panel k+1 on the left columns would move: those are queued on
//    the update stream, behind that GEMM.
struct LuStreams {
  hipStream_t panel = nullptr;
  hipStream_t update = nullptr;    // null: the caller's stream does the updates
  hipStream_t update2 = nullptr;   // second update stream (same CUs as `update`): the right part of a split trailing update
  hipEvent_t left_done[2] = {nullptr, nullptr};   // update  -> update2: left part of step k (parity k & 1) updated
  hipEvent_t right_done[2] = {nullptr, nullptr};  // update2 -> update, panel: right part of step k updated, scratch of step k free
  hipEvent_t resplit = nullptr;    // update -> update2: everything queued before the columns were dealt anew
  hipEvent_t slice_done = nullptr; // panel -> update streams: the panel stream's own slice of the trailing update
  hipEvent_t ready = nullptr;      // update -> panel: next panel's columns are current
  hipEvent_t factored = nullptr;   // panel -> update: panel factorised, transpositions known
  hipEvent_t inverted = nullptr;   // panel -> update: the inverted triangles of the step are there
  hipEvent_t fork = nullptr;       // caller's stream -> update stream
  hipEvent_t join = nullptr;       // update stream -> caller's stream
  int panel_cus = 0;               // CUs set aside for the panel stream (0: none, streams share the chip)
};

// Width of an outer step of the look-ahead.  The block-size rule of the reference (at most
// 256) fixes how a panel is factorised; the TRAILING update may take several such panels at
// once: a step of width W factorises its W columns with blocks of 256 (blocked(), on the panel
// stream: the 256-wide panels, their transpositions, solves and products inside the W columns)
// and then updates the columns to the right with ONE product of reduction length W.  The
// arithmetic per element is the same sum of products, accumulated in one launch instead of
// W / 256; the trailing matrix is read and written W / 256 times less often, which is what
// bounds a k = 256 product (55 TFLOP/s at 32768^2 against 67 at k = 1024).
// While the trailing matrix is large the step is bounded by the trailing product and wide steps
// pay; once the panel (a chain of latency-bound kernels on its own few compute units) is what
// the update stream waits for, wide steps only lengthen that chain, so the width falls back to
// one panel per step.  `remaining` = columns from the step's first column to the end.
// EIGAMD_LU_WIDE fixes the width (a multiple of 256 up to kMaxSub * 256; 256 = one panel per
// step); EIGAMD_LU_WIDE_T4 / _T2 move the thresholds.
struct WidthRule {
  int64_t fixed = 0, t4 = 20480, t2 = 12288;   // (t2 swept again with the faster base cases: 16384^2 114.0 -> 109.0 ms against 16640)
};
// `single`: float, whose trailing products take half the time while its panels take the same: two-panel steps pay later
// (16640: sgetrf 14336^2 73.9 ms against 75.4 with 12288; the complex kinds follow double: zgetrf 16384^2 294.8 -> 284.2).
WidthRule read_width_rule(bool single) {  // (per factorisation: tests set the variables between calls)
  WidthRule v;
  if (single) v.t2 = 16640;
  if (const char* e = getenv("EIGAMD_LU_WIDE")) v.fixed = atoll(e);
  if (const char* e = getenv("EIGAMD_LU_WIDE_T4")) v.t4 = atoll(e);
  if (const char* e = getenv("EIGAMD_LU_WIDE_T2")) v.t2 = atoll(e);
  return v;
}
int64_t outer_block(const WidthRule& r, int64_t remaining, int64_t rule_block) {
  if (rule_block < kPlanSpan) return rule_block;  // small matrices: the reference's own block
  int64_t w = r.fixed > 0 ? r.fixed : (remaining >= r.t4 ? 4 * kPlanSpan : (remaining >= r.t2 ? 2 * kPlanSpan : kPlanSpan));
  return std::max<int64_t>(kPlanSpan, std::min<int64_t>(w / kPlanSpan * kPlanSpan, (int64_t)kMaxSub * kPlanSpan));
}

template <typename T>
hipError_t blocked_lookahead(T* A, int64_t ld, int64_t rows, int64_t cols, int* transp, int64_t max_block,
                             const Workspace<T>& w, const LuStreams& st, hipStream_t upd) {
  const int64_t size = std::min(rows, cols);
  if (size <= kMaxBase) return blocked<T>(A, ld, 0, 0, rows, cols, transp, max_block, w, upd);
  const int64_t rule_block = block_size_rule(size, max_block);
  const WidthRule wr = {w.wide_fixed, w.wide_t4, w.wide_t2};
  hipStream_t pan = st.panel;
  hipStream_t caller = upd;
  if (st.update) {  // the updates run on the stream that leaves the panel's CUs alone
    LU_TRY(hipEventRecord(st.fork, caller));
    LU_TRY(hipStreamWaitEvent(st.update, st.fork, 0));
    upd = st.update;
  }
  LU_TRY(hipEventRecord(st.ready, upd));  // everything queued before the call
  // The bulk of a step's trailing update may be dealt over two streams that share the update CUs: `upd`
  // takes the columns [lo, mid), `upd2` the columns [mid, size).  Each part is the same chain as before
  // (transpositions, solve, product) on its own columns, so nothing changes per element; what changes is
  // that the memory-bound transpositions and the few-workgroup solve of one part run beside the product
  // of the other instead of in front of it.  `mid` stays where it is from step to step (the right part
  // never changes hands, the left part shrinks) until the left part has become small; then the columns are
  // dealt anew, `upd` waiting for the right part's last step before it touches any of those columns.
  hipStream_t upd2 = (st.update && st.update2 && w.split_min > 0) ? st.update2 : nullptr;
  int64_t mid = -1;                          // first column of the right part; -1: the update is not split
  bool right_recorded[2] = {false, false};   // right_done[parity] has been recorded in this call
  // While the trailing matrix is large the panel stream's compute units idle for a good part of every step
  // (the step lasts as long as the trailing product).  They then update the LAST slice_cols columns of the
  // trailing matrix themselves -- the same chain on those columns, queued behind the step's panel -- and the
  // update streams leave those columns alone, from the first step until fewer than slice_min columns remain.
  bool slicing = st.update && w.slice_cols > 0 && size - w.slice_cols >= std::max(w.slice_from, w.slice_min);
  bool slice_recorded = false;
  int step = 0;
  // Width of step `i` when `remaining` columns are left.  The first steps ramp up (one panel, two panels,
  // then the rule's width): nothing runs beside the very first panel, so it should be short.
  auto width_at = [&](int i, int64_t remaining) {
    int64_t wdt = outer_block(wr, remaining, rule_block);
    if (w.ramp && wr.fixed == 0 && i < 2) wdt = std::min<int64_t>(wdt, (int64_t)kPlanSpan << i);
    return std::min(remaining, wdt);
  };
  for (int64_t k = 0, bs = 0; k < size; k += bs, ++step) {
    bs = width_at(step, size - k);
    const int64_t trows = rows - k - bs, tsize = size - k - bs;
    const int par = step & 1;
    // The step's block in pieces of at most kPlanSpan rows: one swap plan and one inverted
    // triangle per piece.  Per-step scratch alternates: the panel stream prepares step k+1
    // while the update streams still use what belongs to step k.
    const int nsub = (int)((bs + kPlanSpan - 1) / kPlanSpan);
    SwapPlan* const* plan = &w.plan[par * kMaxSub];
    T* const* linv = &w.linv[par * kMaxSub];
    int* const* linv_flag = &w.linv_flag[par * kMaxSub];
    auto sub_begin = [&](int i) { return (int64_t)i * kPlanSpan; };
    auto sub_size = [&](int i) { return std::min<int64_t>(kPlanSpan, bs - sub_begin(i)); };
    const bool planned = bs > 32;
    LU_TRY(hipStreamWaitEvent(pan, st.ready, 0));
    if (right_recorded[par]) LU_TRY(hipStreamWaitEvent(pan, st.right_done[par], 0));  // (this parity's scratch: step - 2)
    // The inverted triangles of the step's pieces are built while the pieces are factorised (InvJob), straight into
    // this step's scratch; EIGAMD_LU_INVERSE_ON_THE_WAY=0: by the chain of launches behind the panel, as before.
    InverseTarget<T> targets[kMaxSub];
    for (int i = 0; i < nsub; ++i) targets[i] = {linv[i], linv_flag[i], w.linv_max[par * kMaxSub + i], w.linvT[par * kMaxSub + i], plan[i]};
    const bool on_the_way = incremental_inverse_enabled() && trows > 0 && planned;
    // (pieces_inverted: the plans and inverses of ALL the step's pieces come out of this call)
    bool pieces_inverted;
    if (bs > rule_block) {
      pieces_inverted = on_the_way && rule_block == kPlanSpan && inverse_applies<T>(rule_block);
      LU_TRY(blocked<T>(A, ld, k, k, trows + bs, bs, transp, 16, w, pan, rule_block, pieces_inverted ? targets : nullptr));
    } else {
      // (one piece: this call is the panel level itself)
      pieces_inverted = on_the_way && inverse_applies<T>(bs);
      LU_TRY(blocked<T>(A, ld, k, k, trows + bs, bs, transp, 16, w, pan, 0, pieces_inverted ? targets : nullptr));
    }
    if (planned && !pieces_inverted)   // (otherwise the plans came with the inverses)
      for (int i = 0; i < nsub; ++i)
        LU_TRY(make_plan(transp, (int)(k + sub_begin(i)), (int)(k + sub_begin(i) + sub_size(i)), plan[i], pan));
    // The update streams start on the row exchanges as soon as the plans are there; the inverted triangles,
    // which only the solves need, are worked out beside them (0.16 ms per 256-wide piece that used to sit between
    // the panel and everything that follows it).
    LU_TRY(hipEventRecord(st.factored, pan));
    LU_TRY(hipStreamWaitEvent(upd, st.factored, 0));
    if (trows > 0 && !pieces_inverted)
      for (int i = 0; i < nsub; ++i)
        if (inverse_applies<T>(sub_size(i)))
          LU_TRY(make_inverse<T>(A + (k + sub_begin(i)) * (ld + 1), ld, sub_size(i), linv[i], linv_flag[i], w.linv_max[par * kMaxSub + i], w.inverse_limit, pan));
    LU_TRY(hipEventRecord(st.inverted, pan));
    // the step's transpositions, in order, on `ncols` columns
    auto swap_rows = [&](T* cols, int64_t ncols, hipStream_t q) -> hipError_t {
      if (!planned) return laswp<T>(cols, ld, ncols, transp, (int)k, (int)(k + bs), nullptr, q);
      for (int i = 0; i < nsub; ++i) LU_TRY(apply_plan<T>(cols, ld, ncols, plan[i], q));
      return hipSuccess;
    };
    // A12 <- L^-1 A12 with L the unit lower triangle of the step's bs x bs block, piece by piece:
    // the piece's own triangle (inverted where that applies), then the rows of the later pieces.
    // With `tmp` (bs x ncols, leading dimension bs) the solved rows are written there and A12 keeps the unsolved ones;
    // the caller multiplies with tmp and copies it over A12 when nothing waits for that any more.
    auto solve = [&](T* A12, int64_t ncols, hipStream_t q, T* tmp = nullptr) -> hipError_t {
      if (q != pan) LU_TRY(hipStreamWaitEvent(q, st.inverted, 0));
      for (int i = 0; i < nsub; ++i) {
        const int64_t o = sub_begin(i), sb = sub_size(i);
        const T* L = A + (k + o) * (ld + 1);
        if (inverse_applies<T>(sb)) LU_TRY(apply_inverse<T>(linv[i], linv_flag[i], L, ld, sb, A12 + o, ld, ncols, q, tmp ? tmp + o : nullptr, bs));
        else LU_TRY(trsm_unit_lower<T>(L, A12 + o, ld, sb, ncols, q));
        if (o + sb < bs) LU_TRY(gemm_minus3<T>(bs - o - sb, ncols, sb, L + sb, ld, tmp ? tmp + o : A12 + o, tmp ? bs : ld, A12 + o + sb, ld, q));
      }
      return hipSuccess;
    };
    bool all_inverted = true;
    for (int i = 0; i < nsub; ++i) all_inverted = all_inverted && inverse_applies<T>(sub_size(i));
    T* A21 = A + (k + bs) + k * ld;
    const int64_t nb = (trows && tsize > 0) ? width_at(step + 1, tsize) : 0;
    // how the columns [lo, size) right of the next step's block are dealt in this step
    const int64_t lo = k + bs + nb;
    if (slicing && !(planned && trows && size - w.slice_cols - lo >= w.slice_min)) {
      // the slice goes back to the update streams
      slicing = false;
      if (slice_recorded) {
        LU_TRY(hipStreamWaitEvent(upd, st.slice_done, 0));
        if (upd2) LU_TRY(hipStreamWaitEvent(upd2, st.slice_done, 0));
      }
    }
    const int64_t hi = slicing ? size - w.slice_cols : size;  // the update streams' columns end here
    const int64_t rest = trows ? hi - lo : 0;
    const bool split = upd2 != nullptr && planned && rest >= w.split_min;
    bool dealt_anew = false;
    if (split ? (mid < 0 || mid - lo < (hi - mid) / 3) : mid >= 0) {
      // `upd` is about to touch columns that were the right part's (or stops leaving them alone)
      if (step > 0 && right_recorded[par ^ 1]) LU_TRY(hipStreamWaitEvent(upd, st.right_done[par ^ 1], 0));
      mid = split ? lo + rest / 2 / kPlanSpan * kPlanSpan : -1;
      dealt_anew = split;
    }
    // columns of the next step's block first
    if (nb > 0) {
      // These few columns stand between this panel and the next one.  In place, the product with an inverted triangle
      // is bound to the large tile (a workgroup may only overwrite what it alone has read): two tiles for 256 columns,
      // 70 us on two compute units.  So the solved rows go to scratch, the update reads them there, and they are
      // copied over A12 behind the event the panel stream waits for.
      T* A12 = A + k + (k + bs) * ld;
      T* tmp = (all_inverted && solve_out_of_place_enabled()) ? w.solve_tmp : nullptr;
      LU_TRY(swap_rows(A + (k + bs) * ld, nb, upd));
      LU_TRY(solve(A12, nb, upd, tmp));
      LU_TRY(gemm_minus3<T>(trows, nb, bs, A21, ld, tmp ? tmp : A12, tmp ? bs : ld, A12 + bs, ld, upd));
      LU_TRY(hipEventRecord(st.ready, upd));
      if (tmp) LU_TRY(hipMemcpy2DAsync(A12, (size_t)ld * sizeof(T), tmp, (size_t)bs * sizeof(T), (size_t)bs * sizeof(T), (size_t)nb,
                                       hipMemcpyDeviceToDevice, upd));
    } else {
      LU_TRY(hipEventRecord(st.ready, upd));
    }
    if (dealt_anew) {
      LU_TRY(hipEventRecord(st.resplit, upd));
      LU_TRY(hipStreamWaitEvent(upd2, st.resplit, 0));
    }
    auto bulk = [&](int64_t c0, int64_t c1, hipStream_t q) -> hipError_t {  // trailing columns [c0, c1)
      if (c1 <= c0) return hipSuccess;
      T* A12 = A + k + c0 * ld;
      LU_TRY(swap_rows(A + c0 * ld, c1 - c0, q));
      LU_TRY(solve(A12, c1 - c0, q));
      return gemm_minus<T>(trows, c1 - c0, bs, A21, A12, A12 + bs, ld, q);
    };
    if (!split) {
      // columns to the left (A_0), then the rest of the trailing matrix (A_2)
      if (slice_recorded) LU_TRY(hipStreamWaitEvent(upd, st.slice_done, 0));  // (the slice's product of the step before reads them)
      LU_TRY(swap_rows(A, k, upd));
      if (trows && rest > 0) LU_TRY(bulk(lo, hi, upd));
    } else {
      LU_TRY(bulk(lo, mid, upd));
      LU_TRY(hipStreamWaitEvent(upd2, st.factored, 0));
      LU_TRY(bulk(mid, hi, upd2));
      // The columns to the left: behind the products of the previous step on ALL streams (they read the
      // previous block's columns, which these transpositions move), and in step order -- hence here.
      if (step > 0) LU_TRY(hipStreamWaitEvent(upd2, st.left_done[par ^ 1], 0));
      if (slice_recorded) LU_TRY(hipStreamWaitEvent(upd2, st.slice_done, 0));
      LU_TRY(swap_rows(A, k, upd2));
      LU_TRY(hipEventRecord(st.right_done[par], upd2));
      right_recorded[par] = true;
    }
    LU_TRY(hipEventRecord(st.left_done[par], upd));
    if (slicing) {  // the panel stream's share, behind this step's panel and in front of the next one
      LU_TRY(bulk(hi, size, pan));
      LU_TRY(hipEventRecord(st.slice_done, pan));
      slice_recorded = true;
    }
  }
  if (slicing && slice_recorded) LU_TRY(hipStreamWaitEvent(upd, st.slice_done, 0));
  for (int p = 0; p < 2; ++p)
    if (right_recorded[p]) LU_TRY(hipStreamWaitEvent(upd, st.right_done[p], 0));
  // the panel stream has nothing queued after the last `factored` event
  if (st.update) {
    LU_TRY(hipEventRecord(st.join, upd));
    LU_TRY(hipStreamWaitEvent(caller, st.join, 0));
  }
  return hipSuccess;
}

struct LuScratch {
  std::mutex mu;
  void* buf = nullptr;
  size_t cap = 0;
  unsigned generation = 2;  // pivot columns handed to lu_base_fused so far (never 0: fresh memory reads 0)
  int kind = -1;            // element type the exchange area was last used with
  void* pub_plan = nullptr; // plan scratch of laswp_device
  hipEvent_t pub_plan_done = nullptr;
  // Behind the last factorisation that used `buf`.  A call that does not wait (d_info) returns with its kernels
  // still queued, and the header, plans, inverses and exchange area belong to them until they have run: the next
  // factorisation on this device, from whatever stream or thread, waits for this event first.
  hipEvent_t buf_done = nullptr;
  bool buf_busy = false;    // buf_done has been recorded by a call that did not wait
  LuStreams streams;
};
LuScratch& scratch() {
  static LuScratch s[kMaxDevices];
  return s[current_device_slot()];
}

void release_lu_streams() {
  int count = 0, prev = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return;
  (void)hipGetDevice(&prev);
  for (int d = 0; d < count && d < kMaxDevices; ++d) {
    if (hipSetDevice(d) != hipSuccess) continue;
    LuScratch& sc = scratch();
    std::lock_guard<std::mutex> lock(sc.mu);
    LuStreams& st = sc.streams;
    for (hipStream_t* q : {&st.panel, &st.update, &st.update2})
      if (*q) {
        (void)hipStreamSynchronize(*q);
        (void)hipStreamDestroy(*q);
        *q = nullptr;
      }
    for (hipEvent_t* e : {&st.ready, &st.factored, &st.inverted, &st.fork, &st.join, &st.resplit, &st.left_done[0], &st.left_done[1],
                          &st.right_done[0], &st.right_done[1], &st.slice_done})
      if (*e) {
        (void)hipEventDestroy(*e);
        *e = nullptr;
      }
    st.panel_cus = 0;
  }
  (void)hipSetDevice(prev);
  (void)hipGetLastError();
}

// Streams of the look-ahead.  When the chip has CUs to spare (EIGAMD_LU_PANEL_CUS, default
// 32 of 256; 0 switches this off) the panel stream owns `panel_cus` compute units and the
// update stream the others, so the many short kernels of a panel never queue behind the
// workgroups of the trailing GEMM.  Bit b of a CU mask is compute unit b / 8 of XCD b % 8, so
// a run of bits takes the same number of CUs from every XCD.  The fused base case needs all
// its workgroups resident (`per_cu` fit on a CU): a taller panel gets more CUs.
hipError_t make_streams(LuStreams& st, int base_workgroups, int per_cu, int* total_cus, int default_cus) {
  int want = default_cus;
  if (const char* e = getenv("EIGAMD_LU_PANEL_CUS")) want = atoi(e);
  hipDeviceProp_t prop;
  int dev = 0;
  LU_TRY(hipGetDevice(&dev));
  LU_TRY(hipGetDeviceProperties(&prop, dev));
  const int total = prop.multiProcessorCount;
  *total_cus = total;
  if (want > 0) {
    want = std::max(want, (base_workgroups + per_cu - 1) / per_cu);
    // In steps of 32: four compute units of every XCD, one per shader engine.  The dispatcher deals a launch's workgroups
    // over the shader engines evenly whatever the mask says, so a mask that gives the engines unequal numbers of compute
    // units leaves the fullest engine's last workgroups waiting for a place -- and the one-launch base case needs them all
    // resident: with 48 units (6 per XCD) 96 workgroups of two per unit never came together (complex LU at 24576 rows),
    // with 32 or 64 units an exact fit always has.
    want = (want + 31) / 32 * 32;
    if (total % 32 != 0 || want * 2 > total) want = 0;  // not worth splitting: share the chip
  }
  if (st.panel && st.panel_cus == want) return hipSuccess;
  if (st.panel) {
    LU_TRY(hipStreamSynchronize(st.panel));
    LU_TRY(hipStreamDestroy(st.panel));
    st.panel = nullptr;

    for (hipStream_t* q : {&st.update, &st.update2})
      if (*q) {
        LU_TRY(hipStreamSynchronize(*q));
        LU_TRY(hipStreamDestroy(*q));
        *q = nullptr;
      }
  }
  if (!st.ready) {
    for (hipEvent_t* e : {&st.resplit, &st.left_done[0], &st.left_done[1], &st.right_done[0], &st.right_done[1], &st.slice_done})
      LU_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    LU_TRY(hipEventCreateWithFlags(&st.ready, hipEventDisableTiming));
    LU_TRY(hipEventCreateWithFlags(&st.factored, hipEventDisableTiming));
    LU_TRY(hipEventCreateWithFlags(&st.inverted, hipEventDisableTiming));
    LU_TRY(hipEventCreateWithFlags(&st.fork, hipEventDisableTiming));
    LU_TRY(hipEventCreateWithFlags(&st.join, hipEventDisableTiming));
  }
  if (want == 0) {
    int lo = 0, hi = 0;
    LU_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    LU_TRY(hipStreamCreateWithPriority(&st.panel, hipStreamNonBlocking, hi));
  } else {
    const int words = total / 32;
    std::vector<uint32_t> pmask(words, 0u), umask(words, 0u);
    for (int b = 0; b < total; ++b) (b >= total - want ? pmask : umask)[b / 32] |= 1u << (b % 32);
    LU_TRY(hipExtStreamCreateWithCUMask(&st.panel, (uint32_t)words, pmask.data()));
    LU_TRY(hipExtStreamCreateWithCUMask(&st.update, (uint32_t)words, umask.data()));
    LU_TRY(hipExtStreamCreateWithCUMask(&st.update2, (uint32_t)words, umask.data()));
  }
  st.panel_cus = want;
  // The streams with a compute-unit mask are given back before the process's libraries shut down.  (Left to
  // the end they outlive the runtime objects a profiler keeps about them: under `rocprofv3 --kernel-trace` a
  // Python process that had factorised anything ended in a fault inside the profiler's finaliser; processes
  // that only multiplied or solved, which never create such a stream, did not.)
  static const int registered = std::atexit(release_lu_streams);
  (void)registered;
  if (getenv("EIGAMD_LU_DEBUG")) fprintf(stderr, "eigen_amd LU: panel CUs %d of %d, base-case workgroups %d, %d per CU\n", want, total, base_workgroups, per_cu);
  return hipSuccess;
}

template <typename T>
hipError_t getrf_device_t(int64_t m, int64_t n, T* A, int64_t ld, int* d_transp, int64_t max_block,
                          int* first_zero_pivot, int* nb_transpositions, hipStream_t s, bool allow_fused, int* d_info) {
  LuScratch& sc = scratch();
  std::lock_guard<std::mutex> lock(sc.mu);
  const size_t nparts = (size_t)((m + kRowsPerBlock - 1) / kRowsPerBlock) + 1;
  auto round64 = [](size_t b) { return (b + 63) / 64 * 64; };
  // Layout of the scratch buffer.  Everything the fused base case exchanges through lies at
  // offsets that do not depend on the matrix: a word of the exchange area only ever holds
  // (payload, generation) pairs, so a stale one can never show the current generation.
  const size_t off_side = 256;  // first_zero, nswaps, error, the inverse flags and maxima live in the first 256 bytes
  const size_t off_tops = off_side + round64(sizeof(Side<T>));
  const size_t off_slots = (off_tops + round64(2 * sizeof(BaseTop<T>)) + 1023) / 1024 * 1024;
  const size_t off_plan = off_slots + 2 * (size_t)kMaxFusedParts * sizeof(BaseSlot<T>);
  constexpr int kSets = 2 * kMaxSub + 1;
  const size_t off_linv = off_plan + kSets * round64(sizeof(SwapPlan));
  const size_t off_linvT = off_linv + kSets * round64((size_t)kPlanSpan * kPlanSpan * sizeof(T));
  const size_t off_solve = off_linvT + kSets * round64((size_t)kPlanSpan * kPlanSpan * sizeof(T));
  const size_t off_partials = off_solve + round64((size_t)kMaxSub * kPlanSpan * kMaxSub * kPlanSpan * sizeof(T));
  const size_t need = off_partials + round64(nparts * sizeof(Candidate<Real<T>>));
  if (!sc.buf_done) LU_TRY(hipEventCreateWithFlags(&sc.buf_done, hipEventDisableTiming));
  if (sc.buf_busy) {
    if (need > sc.cap) LU_TRY(hipEventSynchronize(sc.buf_done));  // (the buffer is about to be freed)
    else LU_TRY(hipStreamWaitEvent(s, sc.buf_done, 0));
    sc.buf_busy = false;
  }
  if (need > sc.cap) {
    if (sc.buf) LU_TRY(hipFree(sc.buf));
    sc.buf = nullptr;
    sc.cap = 0;
    LU_TRY(hipMalloc(&sc.buf, need + 4096));
    sc.cap = need + 4096;
    sc.kind = -1;
  }
  // The exchange area is reused from launch to launch without being cleared.  Clear it when
  // the element type (hence the meaning of a slot's words) changes, after a reallocation and
  // before the generation count could wrap.
  if (sc.kind != KindOf<T>::value || sc.generation > 0xF0000000u) {
    LU_TRY(hipMemsetAsync(sc.buf, 0, sc.cap, s));
    sc.kind = KindOf<T>::value;
    sc.generation = 2;
  }
  char* p = static_cast<char*>(sc.buf);
  Workspace<T> w;
  w.rule = read_base_case_rule();
  if (!allow_fused) w.rule.fused = false;
  {
    const WidthRule wr = read_width_rule(std::is_same<T, float>::value);
    w.wide_fixed = wr.fixed; w.wide_t4 = wr.t4; w.wide_t2 = wr.t2;
    w.split_min = 8192;
    w.ramp = true;
    w.slice_cols = 1024;
    // (Swept again with the faster base cases: the slice pays from 25600 columns on -- at 24576 it costs 1 % -- and then
    // until fewer than 20480 remain: 32768^2 489.5 -> 484 ms with the width rule below.)
    w.slice_from = 24576;
    w.slice_min = 20480;
    if (const char* e = getenv("EIGAMD_LU_SLICE")) w.slice_cols = atoll(e) / kPlanSpan * kPlanSpan;
    if (const char* e = getenv("EIGAMD_LU_SLICE_MIN")) w.slice_min = w.slice_from = atoll(e);
    if (const char* e = getenv("EIGAMD_LU_SLICE_FROM")) w.slice_from = atoll(e);
    if (const char* e = getenv("EIGAMD_LU_RAMP")) w.ramp = atoi(e) != 0;
    if (const char* e = getenv("EIGAMD_LU_SPLIT_MIN")) w.split_min = atoll(e);
  }
  w.first_zero = reinterpret_cast<int*>(p);
  w.nswaps = w.first_zero + 1;
  w.error = reinterpret_cast<int*>(p) + 4;
  w.generation = &sc.generation;
  w.side = reinterpret_cast<Side<T>*>(p + off_side);
  w.tops = reinterpret_cast<BaseTop<T>*>(p + off_tops);
  for (int i = 0; i < kSets; ++i) {
    w.plan[i] = reinterpret_cast<SwapPlan*>(p + off_plan + i * round64(sizeof(SwapPlan)));
    w.linv[i] = reinterpret_cast<T*>(p + off_linv + i * round64((size_t)kPlanSpan * kPlanSpan * sizeof(T)));
    w.linvT[i] = reinterpret_cast<T*>(p + off_linvT + i * round64((size_t)kPlanSpan * kPlanSpan * sizeof(T)));
    w.linv_flag[i] = reinterpret_cast<int*>(p) + 8 + i;
    w.linv_max[i] = reinterpret_cast<unsigned long long*>(p + 128) + i;
  }
  static_assert(8 + kSets <= 32 && kSets * 8 <= 128, "the inverse flags and maxima must fit the 256-byte header");
  w.inverse_limit = kInverseGrowthLimit;
  if (const char* e = getenv("EIGAMD_LU_INVERSE_LIMIT")) w.inverse_limit = atof(e);  // 0: always substitute (tests)
  {
    static const bool fuse = [] { const char* e = getenv("EIGAMD_LU_FUSED_ROWOPS"); return !(e && atoi(e) == 0); }();
    w.rank_count = fuse ? reinterpret_cast<int*>(p) + 52 : nullptr;   // bytes [208, 240) of the header, cleared below
    w.rank_fused_max = 4608;
    if (const char* e = getenv("EIGAMD_LU_FUSED_ROWOPS_MAX")) w.rank_fused_max = atoll(e);
    w.far_rides_max = 12288;   // (swept: 8192^2 44.3 ms from 8192 rows on, 16384^2 121.7 / 121.1 / 120.0 / 120.1 and 32768^2 497.5 / 496.1 / 494.7 / 497.9 ms for 4608 / 8192 / 12288 / 16384)
    if (const char* e = getenv("EIGAMD_LU_FAR_RIDES_MAX")) w.far_rides_max = atoll(e);
  }
  static_assert(128 + kSets * 8 <= 208 && 208 + kRankCounters * 4 <= 256 && (kPlanSpan + kRankCols - 1) / kRankCols <= kRankCounters,
                "the counters of lu_rowops_rank_update must fit the header and cover a panel of the look-ahead");
  w.solve_tmp = reinterpret_cast<T*>(p + off_solve);
  w.partials = reinterpret_cast<Candidate<Real<T>>*>(p + off_partials);
  w.slots = reinterpret_cast<BaseSlot<T>*>(p + off_slots);
  LU_TRY(hipMemsetAsync(p, 0, 256, s));  // error flag of the fused base case, inverse flags and maxima
  LU_TRY(hipMemsetAsync(w.first_zero, 0xFF, sizeof(int), s));  // -1
  LU_TRY(hipMemsetAsync(w.nswaps, 0, sizeof(int), s));
  int per_cu = 0, total_cus = 0;
  // Up to some 20000 columns the factorisation waits for the panel chain most of the time, and a panel of up to 64
  // workgroups runs faster with a compute unit per workgroup (first measured as 8192^2 50.6 -> 48.5 ms, 12288^2 84.7 -> 80.0,
  // 16384^2 126.9 -> 124.9 with 64 instead of 32 compute units, and 20480^2 183 -> 188: the trailing products needed them).
  // (The streams are rebuilt when the split changes from one call to the next: a few hardware queues serve all the
  // streams of a process, and a second set kept alive would share them with the first.)
  // (Measured again with the base cases that read every candidate's row beside the poll, 32 / 64 units: 17408^2 130.0 / 122.0 ms, 18432^2 144.8 / 137.2, 20480^2 172.9 / 171.5,
  // 21504^2 194.4 / 190.8, 22528^2 212.4 / 211.8, 24576^2 252.5 / 258.0: 64 up to 22528 columns.)
  const bool panel_bound = std::min(m, n) <= 22528;
  LuStreams& streams = sc.streams;
  LU_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (lu_base_fused<T, false>), kRowsPerBlock, 0));
  {
    int per_cu_wide = 0;
    LU_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_wide, (lu_base_fused<T, true>), kRowsPerBlock, 0));
    per_cu = std::min(per_cu, per_cu_wide);
  }
  if (const char* e = getenv("EIGAMD_LU_WIDE_POLL")) w.wide_poll_min = atoi(e) > 0 ? atoi(e) : kWideParts + 1;
  per_cu = std::max(per_cu, 1);
  if (max_block > 16) {
    LU_TRY(make_streams(streams, (int)(nparts - 1), per_cu, &total_cus, panel_bound ? 64 : 32));
  } else {
    hipDeviceProp_t prop;
    int dev = 0;
    LU_TRY(hipGetDevice(&dev));
    LU_TRY(hipGetDeviceProperties(&prop, dev));
    total_cus = prop.multiProcessorCount;
  }
  const int panel_cus = (max_block > 16 && streams.panel_cus > 0) ? streams.panel_cus : total_cus;
  w.fused_capacity = std::min(kMaxFusedParts, per_cu * panel_cus);
  {
    const int nt = (int)std::min(m, n);
    if (nt > 0) hipLaunchKernelGGL(lu_iota, dim3((nt + 255) / 256), dim3(256), 0, s, d_transp, nt);
    LU_TRY(hipGetLastError());
  }
  if (max_block > 16) {
    LU_TRY(blocked_lookahead<T>(A, ld, m, n, d_transp, max_block, w, streams, s));
  } else {  // a panel: nothing to overlap with
    LU_TRY(blocked<T>(A, ld, 0, 0, m, n, d_transp, max_block, w, s));
  }
  if (d_info) {  // the caller reads the counters from device memory, behind `s`: nothing to wait for
    hipLaunchKernelGGL(lu_export_info, dim3(1), dim3(1), 0, s, w.first_zero, d_info);
    LU_TRY(hipGetLastError());
    LU_TRY(hipEventRecord(sc.buf_done, s));
    sc.buf_busy = true;
    return hipSuccess;
  }
  int out[5] = {-1, 0, 0, 0, 0};
  LU_TRY(hipMemcpyAsync(out, w.first_zero, sizeof out, hipMemcpyDeviceToHost, s));
  LU_TRY(hipStreamSynchronize(s));
  *first_zero_pivot = out[0];
  *nb_transpositions = out[1];
  // A workgroup of the one-launch base case gave up waiting for the others (they were not all
  // resident: the device is shared with other work).  The matrix is then partly factorised and
  // partly garbage; callers that still hold the original run again with allow_fused = false.
  if (out[4] != 0 || (allow_fused && getenv("EIGAMD_LU_INJECT_FUSED_FAILURE"))) return kLuNotResident;
  return hipSuccess;
}

}  // namespace

hipError_t getrf_device(int kind, int64_t m, int64_t n, void* A, int64_t ld, int* d_transp, int64_t max_block,
                        int* first_zero_pivot, int* nb_transpositions, hipStream_t s, bool allow_fused, int* d_info) {
  if (max_block < 8) max_block = 8;
  if (kind == F64)
    return getrf_device_t<double>(m, n, static_cast<double*>(A), ld, d_transp, max_block, first_zero_pivot,
                                  nb_transpositions, s, allow_fused, d_info);
  if (kind == F32)
    return getrf_device_t<float>(m, n, static_cast<float*>(A), ld, d_transp, max_block, first_zero_pivot,
                                 nb_transpositions, s, allow_fused, d_info);
  if (kind == C64)
    return getrf_device_t<Cx<double>>(m, n, static_cast<Cx<double>*>(A), ld, d_transp, max_block, first_zero_pivot,
                                      nb_transpositions, s, allow_fused, d_info);
  if (kind == C32)
    return getrf_device_t<Cx<float>>(m, n, static_cast<Cx<float>*>(A), ld, d_transp, max_block, first_zero_pivot,
                                     nb_transpositions, s, allow_fused, d_info);
  return hipErrorInvalidValue;
}

hipError_t laswp_device(int kind, void* A, int64_t ld, int64_t ncols, const int* d_transp, int k0, int k1,
                        hipStream_t s) {
  if (kind < 0 || kind > 3) return hipErrorInvalidValue;
  // The plan scratch is shared by all callers: a use waits for the previous one to finish.
  LuScratch& sc = scratch();
  std::lock_guard<std::mutex> lock(sc.mu);
  if (!sc.pub_plan) {
    LU_TRY(hipMalloc(&sc.pub_plan, sizeof(SwapPlan)));
    LU_TRY(hipEventCreateWithFlags(&sc.pub_plan_done, hipEventDisableTiming));
  } else {
    LU_TRY(hipStreamWaitEvent(s, sc.pub_plan_done, 0));
  }
  SwapPlan* plan = static_cast<SwapPlan*>(sc.pub_plan);
  hipError_t e = hipSuccess;
  switch (kind) {
    case F64: e = laswp<double>(static_cast<double*>(A), ld, ncols, d_transp, k0, k1, plan, s); break;
    case F32: e = laswp<float>(static_cast<float*>(A), ld, ncols, d_transp, k0, k1, plan, s); break;
    case C64: e = laswp<Cx<double>>(static_cast<Cx<double>*>(A), ld, ncols, d_transp, k0, k1, plan, s); break;
    default:  e = laswp<Cx<float>>(static_cast<Cx<float>*>(A), ld, ncols, d_transp, k0, k1, plan, s); break;
  }
  LU_TRY(e);
  return hipEventRecord(sc.pub_plan_done, s);
}

hipError_t trsm_llnu_device(int kind, int64_t bs, int64_t ncols, const void* L, int64_t ldl, void* B, int64_t ldb,
                            hipStream_t s) {
  if (kind == F64)
    return trsm_unit_lower2<double>(static_cast<const double*>(L), ldl, static_cast<double*>(B), ldb, bs, ncols, s);
  if (kind == F32)
    return trsm_unit_lower2<float>(static_cast<const float*>(L), ldl, static_cast<float*>(B), ldb, bs, ncols, s);
  if (kind == C64)
    return trsm_unit_lower2<Cx<double>>(static_cast<const Cx<double>*>(L), ldl, static_cast<Cx<double>*>(B), ldb, bs, ncols, s);
  if (kind == C32)
    return trsm_unit_lower2<Cx<float>>(static_cast<const Cx<float>*>(L), ldl, static_cast<Cx<float>*>(B), ldb, bs, ncols, s);
  return hipErrorInvalidValue;
}


// ---- ?getrs: the solve that consumes the factors (lapack/lu.cpp:46-89; PartialPivLU::_solve_impl,
// Eigen/src/LU/PartialPivLU.h:227-270) --------------------------------------------------------------

namespace {

struct alignas(8) Pair32 { float re, im; };
struct alignas(16) Pair64 { double re, im; };

// lu_apply_plan for any element type and in either direction: `inverse` undoes the run of
// transpositions the plan stands for (row src receives what row dst held), which is the run applied
// in reverse order.
template <typename T, int COLS>
__global__ void __launch_bounds__(256) lu_apply_plan_dir(T* __restrict__ A, int64_t ld, int64_t ncols,
                                                         const SwapPlan* __restrict__ plan, int inverse) {
  __shared__ T s_val[2 * kPlanSpan * COLS];
  const int n = plan->count;
  const int64_t c0 = (int64_t)blockIdx.x * COLS;
  const int nc = (int)min((int64_t)COLS, ncols - c0);
  const int* from = inverse ? plan->dst : plan->src;
  const int* to = inverse ? plan->src : plan->dst;
  for (int idx = threadIdx.x; idx < n * nc; idx += 256) {
    const int c = idx / n, j = idx - c * n;
    s_val[idx] = A[from[j] + (c0 + c) * ld];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < n * nc; idx += 256) {
    const int c = idx / n, j = idx - c * n;
    A[to[j] + (c0 + c) * ld] = s_val[idx];
  }
}

// The transpositions i <-> transp[i], i in [0, n), on all columns of B: in order (forward) or in
// reverse order, kPlanSpan of them per pair of launches.
template <typename T>
hipError_t permute_rows(T* B, int64_t ldb, int64_t ncols, const int* transp, int64_t n, bool reverse, SwapPlan* plan,
                        hipStream_t s, int64_t first = 0) {
  constexpr int kCols = sizeof(T) > 8 ? 4 : 8;
  const int64_t chunks = (n - first + kPlanSpan - 1) / kPlanSpan;
  for (int64_t q = 0; q < chunks; ++q) {
    const int64_t c = reverse ? chunks - 1 - q : q;
    const int k0 = (int)(first + c * kPlanSpan), k1 = (int)std::min<int64_t>(n, first + (c + 1) * kPlanSpan);
    LU_TRY(make_plan(transp, k0, k1, plan, s));
    hipLaunchKernelGGL((lu_apply_plan_dir<T, kCols>), dim3((unsigned)((ncols + kCols - 1) / kCols)), dim3(256), 0, s, B, ldb,
                       ncols, plan, reverse ? 1 : 0);
    LU_TRY(hipGetLastError());
  }
  return hipSuccess;
}

}  // namespace

hipError_t getrs_device(int kind, int op, int64_t n, int64_t nrhs, const void* LU, int64_t ldlu, const int* d_transp, void* B,
                        int64_t ldb, hipStream_t s) {
  if (kind < 0 || kind > 3 || op < OP_N || op > OP_C) return hipErrorInvalidValue;
  if (n == 0 || nrhs == 0) return hipSuccess;
  LuScratch& sc = scratch();
  std::lock_guard<std::mutex> lock(sc.mu);
  if (!sc.pub_plan) {
    LU_TRY(hipMalloc(&sc.pub_plan, sizeof(SwapPlan)));
    LU_TRY(hipEventCreateWithFlags(&sc.pub_plan_done, hipEventDisableTiming));
  } else {
    LU_TRY(hipStreamWaitEvent(s, sc.pub_plan_done, 0));
  }
  SwapPlan* plan = static_cast<SwapPlan*>(sc.pub_plan);
  auto permute = [&](bool reverse) -> hipError_t {
    switch (kind) {
      case F32: return permute_rows<float>(static_cast<float*>(B), ldb, nrhs, d_transp, n, reverse, plan, s);
      case F64: return permute_rows<double>(static_cast<double*>(B), ldb, nrhs, d_transp, n, reverse, plan, s);
      case C32: return permute_rows<Pair32>(static_cast<Pair32*>(B), ldb, nrhs, d_transp, n, reverse, plan, s);
      default:  return permute_rows<Pair64>(static_cast<Pair64*>(B), ldb, nrhs, d_transp, n, reverse, plan, s);
    }
  };
  const double one[2] = {1.0, 0.0};
  if (op == OP_N) {                                  // lu.cpp:67-72: B = P B, then L, then U
    LU_TRY(permute(false));
    LU_TRY(hipEventRecord(sc.pub_plan_done, s));
    LU_TRY(trsm_device(kind, true, /*upper=*/false, OP_N, /*unit=*/true, n, nrhs, one, LU, ldlu, B, ldb, s));
    LU_TRY(trsm_device(kind, true, /*upper=*/true, OP_N, /*unit=*/false, n, nrhs, one, LU, ldlu, B, ldb, s));
  } else {                                           // lu.cpp:73-84: op(U), then op(L), then B = P^T B
    LU_TRY(trsm_device(kind, true, /*upper=*/true, op, /*unit=*/false, n, nrhs, one, LU, ldlu, B, ldb, s));
    LU_TRY(trsm_device(kind, true, /*upper=*/false, op, /*unit=*/true, n, nrhs, one, LU, ldlu, B, ldb, s));
    LU_TRY(permute(true));
    LU_TRY(hipEventRecord(sc.pub_plan_done, s));
  }
  return hipSuccess;
}

}  // namespace eigamd
