// Blocked LU with partial pivoting for the complex kinds, device-resident.
//
// partial_lu_impl::blocked_lu / unblocked_lu (Eigen/src/LU/PartialPivLU.h:357-498) are templates: the
// reference runs the same algorithm for std::complex scalars (cgetrf / zgetrf bindings,
// PartialPivLU_LAPACKE.h:77-78, lapack/complex_single.cpp, complex_double.cpp).  The pivot of a column is
// the first entry of largest modulus (scalar_score_coeff_op = abs, :375-377).  This driver keeps the
// reference's structure -- block-size rule, panels factorised with blocks of 16, transpositions applied
// left and right in its order, unit-lower solve, trailing update on the complex MFMA GEMM -- with the
// plainest device form of each step: the base case is a chain of three small kernels per column (the
// kernel boundary is the grid-wide synchronisation), everything on the caller's stream, no look-ahead.
// The real kinds have the tuned driver of lu.hip; a complex factorisation is not on the configured path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <mutex>

#include "gemm_launch.hpp"
#include "lu.hpp"
#include "trsm.hpp"

namespace eigamd {

namespace {

constexpr int kRows = 256;   // rows of a panel one workgroup owns
constexpr int kBase = 16;    // widest unblocked panel (PartialPivLU.h:336)
constexpr int kNoRow = 0x7fffffff;

template <typename R> struct Z {
  R x, y;
};
template <typename R> __device__ __forceinline__ Z<R> operator*(Z<R> a, Z<R> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename R> __device__ __forceinline__ Z<R> operator-(Z<R> a, Z<R> b) { return {a.x - b.x, a.y - b.y}; }
// Smith's division, as std::complex does it
template <typename R> __device__ __forceinline__ Z<R> operator/(Z<R> a, Z<R> b) {
  if (fabs(b.x) >= fabs(b.y)) {
    const R r = b.y / b.x, d = b.x + b.y * r;
    return {(a.x + a.y * r) / d, (a.y - a.x * r) / d};
  }
  const R r = b.x / b.y, d = b.x * r + b.y;
  return {(a.x * r + a.y) / d, (a.y * r - a.x) / d};
}
template <typename R> __device__ __forceinline__ R modulus(Z<R> a) { return hypot(a.x, a.y); }

template <typename R> struct Cand {
  R val;    // |a|, or -1 for "no row"
  int row;
};
template <typename R> struct PivotRow {
  Z<R> u[kBase];
  Z<R> pivot;
  int zero;
};
template <typename R> __device__ __forceinline__ bool better(R v, int r, R ov, int orow) { return (v > ov) | ((v == ov) & (r < orow)); }

template <typename R> __device__ Cand<R> block_best(R val, int row) {
  __shared__ R s_val[kRows / 64];
  __shared__ int s_row[kRows / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const R ov = __shfl_down(val, off, 64);
    const int orow = __shfl_down(row, off, 64);
    if (better(ov, orow, val, row)) { val = ov; row = orow; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { s_val[wave] = val; s_row[wave] = row; }
  __syncthreads();
  Cand<R> c = {s_val[0], s_row[0]};
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
    if (better(s_val[w], s_row[w], c.val, c.row)) { c.val = s_val[w]; c.row = s_row[w]; }
  return c;
}

template <typename R>
__global__ void zlu_colmax(const Z<R>* __restrict__ P, int64_t ld, int rows, Cand<R>* __restrict__ partials) {
  const int r = blockIdx.x * kRows + threadIdx.x;
  R val = (R)-1;
  if (r < rows) val = modulus(P[r]);
  const Cand<R> c = block_best<R>(val, r < rows ? r : kNoRow);
  if (threadIdx.x == 0) partials[blockIdx.x] = c;
}

template <typename R>
__global__ void zlu_pivot(Z<R>* __restrict__ P, int64_t ld, int jbw, int jj, const Cand<R>* __restrict__ partials, int nparts,
                          int* __restrict__ transp, int row_offset, int col_global, PivotRow<R>* __restrict__ side,
                          int* __restrict__ first_zero, int* __restrict__ nswaps) {
  const int lane = threadIdx.x;
  R val = (R)-1;
  int row = kNoRow;
  for (int i = lane; i < nparts; i += 64) {
    const Cand<R> c = partials[i];
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
  if (row == kNoRow) {  // nothing comparable in the column (NaNs): no exchange, pivot = the diagonal entry
    row = jj;
    val = (R)1;
  }
  const bool zero = !(val != (R)0);
  if (lane == 0) {
    transp[0] = row_offset + row;
    if (zero) atomicCAS(first_zero, -1, col_global);  // the FIRST zero pivot (PartialPivLU.h:391-396)
    else if (row != jj) atomicAdd(nswaps, 1);
  }
  if (lane < jbw) {
    Z<R> top = P[jj + (int64_t)lane * ld];
    if (!zero && row != jj) {
      const Z<R> other = P[row + (int64_t)lane * ld];
      P[row + (int64_t)lane * ld] = top;
      P[jj + (int64_t)lane * ld] = other;
      top = other;
    }
    side->u[lane] = top;
    if (lane == jj) { side->pivot = top; side->zero = zero ? 1 : 0; }
  }
}

template <typename R>
__global__ void zlu_update(Z<R>* __restrict__ P, int64_t ld, int rows, int jbw, int jj, const PivotRow<R>* __restrict__ side,
                           Cand<R>* __restrict__ partials) {
  __shared__ Z<R> s_u[kBase];
  __shared__ Z<R> s_pivot;
  __shared__ int s_zero;
  if (threadIdx.x < kBase) s_u[threadIdx.x] = side->u[threadIdx.x];
  if (threadIdx.x == 0) { s_pivot = side->pivot; s_zero = side->zero; }
  __syncthreads();
  const int r = jj + 1 + blockIdx.x * kRows + threadIdx.x;
  R cand = (R)-1;
  if (r < rows) {
    Z<R> l = P[r + (int64_t)jj * ld];
    if (!s_zero) {
      l = l / s_pivot;  // the reference divides (lu.col(k).tail /= pivot, :389)
      P[r + (int64_t)jj * ld] = l;
    }
    for (int c = jj + 1; c < jbw; ++c) {
      Z<R>* dst = P + r + (int64_t)c * ld;
      const Z<R> x = *dst - l * s_u[c];
      *dst = x;
      if (c == jj + 1) cand = modulus(x);
    }
  }
  if (jj + 1 < jbw) {
    const Cand<R> c = block_best<R>(cand, r < rows ? r : kNoRow);
    if (threadIdx.x == 0) partials[blockIdx.x] = c;
  }
}

// One pivot column in ONE launch (zlu_pivot and zlu_update together; EIGAMD_ZLU_ONE_LAUNCH=0 keeps the two).  Every
// workgroup picks the pivot from the candidates of the launch before (all pick the same), reads panel rows jj and p
// as they are, and eliminates its own rows with u = row p; the thread that owns row p works on what row jj held
// instead of its own row and stores nothing.  Rows jj and p themselves are written once nobody reads them any more:
// a workgroup counts itself in when its copies of them are in LDS, and the workgroup that completes the count stores
// row jj <- old row p and row p <- eliminated old row jj at its end, and puts the counter back.  Nobody waits.  The
// candidates of the next column go to the other of two buffers.  Same operations per element as the two launches.
template <typename R>
__global__ void __launch_bounds__(kRows) zlu_step(Z<R>* __restrict__ P, int64_t ld, int rows, int jbw, int jj,
                                                  const Cand<R>* __restrict__ cand_in, int nin, Cand<R>* __restrict__ cand_out,
                                                  int* __restrict__ transp, int row_offset, int col_global,
                                                  int* __restrict__ first_zero, int* __restrict__ nswaps, int* __restrict__ counter) {
  __shared__ Z<R> s_top[kBase], s_piv[kBase];  // rows jj and p of the panel before this column's exchange (s_piv = u)
  __shared__ R s_best;
  __shared__ int s_row, s_last;
  const int t = threadIdx.x;
  if (t < 64) {
    R val = (R)-1;
    int row = kNoRow;
    for (int i = t; i < nin; i += 64) {
      const Cand<R> c = cand_in[i];
      if (better(c.val, c.row, val, row)) { val = c.val; row = c.row; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const R ov = __shfl_down(val, off, 64);
      const int orow = __shfl_down(row, off, 64);
      if (better(ov, orow, val, row)) { val = ov; row = orow; }
    }
    if (t == 0) {
      if (row == kNoRow) {  // nothing comparable in the column (NaNs): no exchange, pivot = the diagonal entry
        row = jj;
        val = (R)1;
      }
      s_row = row;
      s_best = val;
    }
  }
  __syncthreads();
  const int p = s_row;
  const bool zero = !(s_best != (R)0);
  const bool swap = !zero && p != jj;
  if (t < jbw) {
    const Z<R> top = P[jj + (int64_t)t * ld];
    s_top[t] = top;
    s_piv[t] = swap ? P[p + (int64_t)t * ld] : top;
  }
  __syncthreads();
  if (t == 0) {  // this workgroup's copies of the two rows are in LDS
    const int arrived = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = arrived == (int)gridDim.x - 1;
    if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last;
    if (blockIdx.x == 0) {
      transp[0] = row_offset + p;
      if (zero) atomicCAS(first_zero, -1, col_global);  // the FIRST zero pivot (PartialPivLU.h:391-396)
      else if (swap) atomicAdd(nswaps, 1);
    }
  }
  const Z<R> pivot = s_piv[jj];
  const int r = jj + 1 + blockIdx.x * kRows + t;
  R cand = (R)-1;
  if (r < rows) {
    const bool is_p = swap && r == p;  // this row receives what row jj held; the last workgroup stores it
    Z<R> l = is_p ? s_top[jj] : P[r + (int64_t)jj * ld];
    if (!zero) {
      l = l / pivot;  // the reference divides (lu.col(k).tail /= pivot, :389)
      if (!is_p) P[r + (int64_t)jj * ld] = l;
    }
    for (int c = jj + 1; c < jbw; ++c) {
      Z<R>* dst = P + r + (int64_t)c * ld;
      const Z<R> x = (is_p ? s_top[c] : *dst) - l * s_piv[c];
      if (!is_p) *dst = x;
      if (c == jj + 1) cand = modulus(x);
    }
  }
  if (jj + 1 < jbw) {
    const Cand<R> c = block_best<R>(cand, r < rows ? r : kNoRow);
    if (t == 0) cand_out[blockIdx.x] = c;
  } else {
    __syncthreads();
  }
  if (s_last && swap && t < jbw) {
    P[jj + (int64_t)t * ld] = s_piv[t];
    Z<R> v = s_top[t];
    if (t >= jj) {
      const Z<R> l = s_top[jj] / pivot;
      v = t == jj ? l : s_top[t] - l * s_piv[t];
    }
    P[p + (int64_t)t * ld] = v;
  }
}

// transpositions [k0, k1) on nleft + nright columns, and on the right-hand ones the forward substitution
// with the unit lower jb x jb triangle L at rows [k0, k0 + jb); one thread per column
template <typename R>
__global__ void zlu_rowops(Z<R>* __restrict__ A_left, int nleft, Z<R>* __restrict__ A_right, int nright, int64_t ld,
                           const int* __restrict__ transp, int k0, int k1, const Z<R>* __restrict__ L, int jb) {
  __shared__ Z<R> s_l[kBase * kBase];
  __shared__ int s_p[kBase];
  for (int i = threadIdx.x; i < jb * jb; i += blockDim.x) s_l[(i % jb) + (i / jb) * kBase] = L[(i % jb) + (int64_t)(i / jb) * ld];
  if ((int)threadIdx.x < k1 - k0) s_p[threadIdx.x] = transp[k0 + threadIdx.x];
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nleft + nright) return;
  const bool right = c >= nleft;
  Z<R>* col = right ? A_right + (int64_t)(c - nleft) * ld : A_left + (int64_t)c * ld;
  for (int i = k0; i < k1; ++i) {
    const int p = s_p[i - k0];
    if (p != i) {
      const Z<R> t = col[i];
      col[i] = col[p];
      col[p] = t;
    }
  }
  if (!right) return;
  for (int i = 0; i < jb; ++i) {
    const Z<R> xi = col[k0 + i];
    for (int rr = i + 1; rr < jb; ++rr) col[k0 + rr] = col[k0 + rr] - s_l[rr + i * kBase] * xi;
  }
}

// C (rows x cols) -= A (rows x jb) * B (jb x cols), jb <= 16; thread = one row of A, 16 columns per workgroup
template <typename R>
__global__ void __launch_bounds__(256) zlu_rank_update(const Z<R>* __restrict__ A, const Z<R>* __restrict__ B, Z<R>* __restrict__ C,
                                                       int64_t ld, int rows, int cols, int jb) {
  constexpr int kCols = 16;
  __shared__ Z<R> s_b[kBase][kCols];
  const int c0 = blockIdx.y * kCols;
  const int nc = min(kCols, cols - c0);
  for (int idx = threadIdx.x; idx < kBase * kCols; idx += 256) {
    const int k = idx / kCols, j = idx % kCols;
    const Z<R> z = {(R)0, (R)0};
    s_b[k][j] = (k < jb && j < nc) ? B[k + (int64_t)(c0 + j) * ld] : z;
  }
  __syncthreads();
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  Z<R> a[kBase];
#pragma unroll
  for (int k = 0; k < kBase; ++k) {
    const Z<R> z = {(R)0, (R)0};
    a[k] = k < jb ? A[r + (int64_t)k * ld] : z;
  }
  // groups of 8 columns: their loads are independent and in flight together (one at a time, the kernel waited for
  // memory sixteen times per row)
  constexpr int kGroup = 8;
  int j = 0;
  for (; j + kGroup <= nc; j += kGroup) {
    Z<R> acc[kGroup];
#pragma unroll
    for (int u = 0; u < kGroup; ++u) acc[u] = C[r + (int64_t)(c0 + j + u) * ld];
#pragma unroll
    for (int u = 0; u < kGroup; ++u) {
#pragma unroll
      for (int k = 0; k < kBase; ++k) acc[u] = acc[u] - a[k] * s_b[k][j + u];
    }
#pragma unroll
    for (int u = 0; u < kGroup; ++u) C[r + (int64_t)(c0 + j + u) * ld] = acc[u];
  }
  for (; j < nc; ++j) {
    Z<R> acc = C[r + (int64_t)(c0 + j) * ld];
#pragma unroll
    for (int k = 0; k < kBase; ++k) acc = acc - a[k] * s_b[k][j];
    C[r + (int64_t)(c0 + j) * ld] = acc;
  }
}

template <typename R>
__global__ void zlu_laswp(Z<R>* __restrict__ A, int64_t ld, int ncols, const int* __restrict__ transp, int k0, int k1) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  Z<R>* col = A + (int64_t)c * ld;
  for (int i = k0; i < k1; ++i) {
    const int p = transp[i];
    if (p != i) {
      const Z<R> t = col[i];
      col[i] = col[p];
      col[p] = t;
    }
  }
}

// B <- L^-1 B for a unit lower jb x jb block (jb <= 32) and jb x ncols B: one thread per column
template <typename R>
__global__ void zlu_trsm_small(const Z<R>* __restrict__ L, int64_t ldl, Z<R>* __restrict__ B, int64_t ldb, int jb, int ncols) {
  constexpr int JB = 32;
  __shared__ Z<R> s_l[JB * JB];
  for (int i = threadIdx.x; i < jb * jb; i += blockDim.x) s_l[(i % jb) + (i / jb) * JB] = L[(i % jb) + (int64_t)(i / jb) * ldl];
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  Z<R>* col = B + (int64_t)c * ldb;
  for (int i = 0; i < jb; ++i) {
    const Z<R> xi = col[i];
    for (int r = i + 1; r < jb; ++r) col[r] = col[r] - s_l[r + i * JB] * xi;
  }
}

__global__ void zlu_iota(int* __restrict__ transp, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) transp[i] = i;
}

#define ZLU_TRY(expr)                  \
  do {                                 \
    hipError_t e_ = (expr);            \
    if (e_ != hipSuccess) return e_;   \
  } while (0)

template <typename R> struct ZKind;
template <> struct ZKind<float> { static constexpr int value = C32; };
template <> struct ZKind<double> { static constexpr int value = C64; };

template <typename R> hipError_t zgemm_minus(int64_t m, int64_t n, int64_t k, const Z<R>* A, const Z<R>* B, Z<R>* C, int64_t ld, hipStream_t s) {
  if (m <= 0 || n <= 0 || k <= 0) return hipSuccess;
  GemmCall c;
  c.kind = ZKind<R>::value;
  c.opa = OP_N; c.opb = OP_N;
  c.m = m; c.n = n; c.k = k;
  c.alpha[0] = -1.0; c.alpha[1] = 0.0;
  c.beta[0] = 1.0; c.beta[1] = 0.0;
  c.A = A; c.lda = ld;
  c.B = B; c.ldb = ld;
  c.C = C; c.ldc = ld;
  return ZKind<R>::value == C64 ? launch_gemm_c64(c, s) : launch_gemm_c32(c, s);
}

template <typename R> struct ZWork {
  Cand<R>* partials2;   // the second candidate buffer of zlu_step
  int* count;           // its counter (zero between launches), or null: two launches per pivot column
  Cand<R>* partials;
  PivotRow<R>* side;
  int* first_zero;
  int* nswaps;
};

template <typename R>
hipError_t zunblocked(Z<R>* P, int64_t ld, int64_t rows, int jbw, int* transp, int row_offset, int col_offset, const ZWork<R>& w,
                      hipStream_t s) {
  const int size = (int)std::min<int64_t>(rows, jbw);
  int nparts = (int)((rows + kRows - 1) / kRows);
  hipLaunchKernelGGL(zlu_colmax<R>, dim3(nparts), dim3(kRows), 0, s, P, ld, (int)rows, w.partials);
  if (w.count) {
    Cand<R>* buf[2] = {w.partials, w.partials2};
    for (int jj = 0; jj < size; ++jj) {
      const int64_t below = rows - jj - 1;
      const int nout = (int)std::max<int64_t>(1, (below + kRows - 1) / kRows);
      hipLaunchKernelGGL(zlu_step<R>, dim3(nout), dim3(kRows), 0, s, P, ld, (int)rows, jbw, jj, buf[jj & 1], nparts, buf[(jj + 1) & 1],
                         transp + jj, row_offset, col_offset + jj, w.first_zero, w.nswaps, w.count);
      nparts = nout;
    }
    return hipGetLastError();
  }
  for (int jj = 0; jj < size; ++jj) {
    hipLaunchKernelGGL(zlu_pivot<R>, dim3(1), dim3(64), 0, s, P, ld, jbw, jj, w.partials, nparts, transp + jj, row_offset,
                       col_offset + jj, w.side, w.first_zero, w.nswaps);
    const int64_t below = rows - jj - 1;
    if (below > 0) {
      nparts = (int)((below + kRows - 1) / kRows);
      hipLaunchKernelGGL(zlu_update<R>, dim3(nparts), dim3(kRows), 0, s, P, ld, (int)rows, jbw, jj, w.side, w.partials);
    }
  }
  return hipGetLastError();
}

template <typename R>
hipError_t zlaswp(Z<R>* A, int64_t ld, int64_t ncols, const int* transp, int k0, int k1, hipStream_t s) {
  if (ncols <= 0 || k1 <= k0) return hipSuccess;
  // long runs: as permutations of 256 transpositions with independent loads (lu.hip) instead of a chain of dependent
  // exchanges per column (256 of them: 0.25 ms whatever the number of columns)
  if (k1 - k0 > 32) return laswp_device_complex(ZKind<R>::value, A, ld, ncols, transp, k0, k1, s);
  hipLaunchKernelGGL(zlu_laswp<R>, dim3((unsigned)((ncols + 127) / 128)), dim3(128), 0, s, A, ld, (int)ncols, transp, k0, k1);
  return hipGetLastError();
}

int block_size_rule(int64_t size, int64_t max_block) {  // PartialPivLU.h:443-448
  int64_t b = size / 8;
  b = (b / 16) * 16;
  return (int)std::min<int64_t>(std::max<int64_t>(b, 8), max_block);
}

// blocked_lu (PartialPivLU.h:429-498); transposition entries are GLOBAL row indices of `base`
template <typename R>
hipError_t zblocked(Z<R>* base, int64_t ld, int64_t row0, int64_t col0, int64_t rows, int64_t cols, int* transp, int64_t max_block,
                    const ZWork<R>& w, hipStream_t s) {
  Z<R>* A = base + row0 + col0 * ld;
  const int64_t size = std::min(rows, cols);
  if (size <= kBase) {
    const int jbw = (int)std::min<int64_t>(cols, rows);
    ZLU_TRY(zunblocked<R>(A, ld, rows, jbw, transp + row0, (int)row0, (int)col0, w, s));
    if (cols > jbw) {  // wider than tall: the same operations on the columns to the right
      ZLU_TRY(zlaswp<R>(base + (col0 + jbw) * ld, ld, cols - jbw, transp, (int)row0, (int)(row0 + jbw), s));
      hipLaunchKernelGGL(zlu_trsm_small<R>, dim3((unsigned)((cols - jbw + 63) / 64)), dim3(64), 0, s, A, ld, A + (int64_t)jbw * ld, ld,
                         jbw, (int)(cols - jbw));
    }
    return hipGetLastError();
  }
  const int64_t block = block_size_rule(size, max_block);
  for (int64_t k = 0; k < size; k += block) {
    const int64_t bs = std::min(size - k, block);
    const int64_t trows = rows - k - bs, tsize = size - k - bs;
    ZLU_TRY(zblocked<R>(base, ld, row0 + k, col0 + k, trows + bs, bs, transp, 16, w, s));  // the panel (:472-473)
    Z<R>* A11 = A + k + k * ld;
    Z<R>* A12 = A + k + (k + bs) * ld;
    Z<R>* A21 = A + (k + bs) + k * ld;
    Z<R>* A22 = A + (k + bs) + (k + bs) * ld;
    if (bs <= kBase) {
      const int64_t nright = trows ? tsize : 0;
      if (k + nright > 0) {
        hipLaunchKernelGGL(zlu_rowops<R>, dim3((unsigned)((k + nright + 63) / 64)), dim3(64), 0, s, base + col0 * ld, (int)k,
                           base + (col0 + k + bs) * ld, (int)nright, ld, transp, (int)(row0 + k), (int)(row0 + k + bs), A11, (int)bs);
        ZLU_TRY(hipGetLastError());
      }
      if (trows && tsize > 0) {
        hipLaunchKernelGGL(zlu_rank_update<R>, dim3((unsigned)((trows + 255) / 256), (unsigned)((tsize + 15) / 16)), dim3(256), 0, s, A21,
                           A12, A22, ld, (int)trows, (int)tsize, (int)bs);
        ZLU_TRY(hipGetLastError());
      }
      continue;
    }
    ZLU_TRY(zlaswp<R>(base + col0 * ld, ld, k, transp, (int)(row0 + k), (int)(row0 + k + bs), s));                 // A_0 (:479-483)
    if (trows && tsize > 0) {
      ZLU_TRY(zlaswp<R>(base + (col0 + k + bs) * ld, ld, tsize, transp, (int)(row0 + k), (int)(row0 + k + bs), s));  // A_2 (:488-489)
      static const bool lib_solve = [] { const char* e = getenv("EIGAMD_ZLU_TRSM"); return !(e && atoi(e) == 0); }();
      if (lib_solve && bs >= 64) {  // the library's triangular solve (inverted 128-wide blocks, guarded; trsm.hip)       // (:492)
        const double one[2] = {1.0, 0.0};
        ZLU_TRY(trsm_device(ZKind<R>::value, true, false, OP_N, true, bs, tsize, one, A11, ld, A12, ld, s));
      } else
      for (int64_t j = 0; j < bs; j += 32) {
        const int jb = (int)std::min<int64_t>(32, bs - j);
        hipLaunchKernelGGL(zlu_trsm_small<R>, dim3((unsigned)((tsize + 63) / 64)), dim3(64), 0, s, A11 + j + j * ld, ld, A12 + j, ld, jb,
                           (int)tsize);
        ZLU_TRY(zgemm_minus<R>(bs - j - jb, tsize, jb, A11 + (j + jb) + j * ld, A12 + j, A12 + j + jb, ld, s));
      }
      ZLU_TRY(zgemm_minus<R>(trows, tsize, bs, A21, A12, A22, ld, s));                                              // (:494)
    }
  }
  return hipSuccess;
}

struct ZScratch {
  std::mutex mu;
  void* buf = nullptr;
  size_t cap = 0;
};
ZScratch& zscratch() {
  static ZScratch s[kMaxDevices];
  return s[current_device_slot()];
}

template <typename R>
hipError_t zgetrf_t(int64_t m, int64_t n, Z<R>* A, int64_t ld, int* d_transp, int64_t max_block, int* first_zero_pivot,
                    int* nb_transpositions, hipStream_t s) {
  ZScratch& sc = zscratch();
  std::lock_guard<std::mutex> lock(sc.mu);
  const size_t nparts = (size_t)((m + kRows - 1) / kRows) + 1;
  const size_t off_side = 64, off_partials = off_side + (sizeof(PivotRow<R>) + 63) / 64 * 64;
  const size_t partial_bytes = (nparts * sizeof(Cand<R>) + 63) / 64 * 64;
  const size_t need = off_partials + 2 * partial_bytes;
  if (need > sc.cap) {
    if (sc.buf) ZLU_TRY(hipFree(sc.buf));
    sc.buf = nullptr; sc.cap = 0;
    ZLU_TRY(hipMalloc(&sc.buf, need + 4096));
    sc.cap = need + 4096;
  }
  char* p = static_cast<char*>(sc.buf);
  ZWork<R> w;
  w.first_zero = reinterpret_cast<int*>(p);
  w.nswaps = w.first_zero + 1;
  w.side = reinterpret_cast<PivotRow<R>*>(p + off_side);
  w.partials = reinterpret_cast<Cand<R>*>(p + off_partials);
  w.partials2 = reinterpret_cast<Cand<R>*>(p + off_partials + partial_bytes);
  static const bool one_launch = [] { const char* e = getenv("EIGAMD_ZLU_ONE_LAUNCH"); return !(e && atoi(e) == 0); }();
  w.count = one_launch ? reinterpret_cast<int*>(p) + 2 : nullptr;
  ZLU_TRY(hipMemsetAsync(p, 0, 64, s));
  ZLU_TRY(hipMemsetAsync(w.first_zero, 0xFF, sizeof(int), s));  // -1
  ZLU_TRY(hipMemsetAsync(w.nswaps, 0, sizeof(int), s));
  const int nt = (int)std::min(m, n);
  if (nt > 0) hipLaunchKernelGGL(zlu_iota, dim3((nt + 255) / 256), dim3(256), 0, s, d_transp, nt);
  ZLU_TRY(hipGetLastError());
  ZLU_TRY(zblocked<R>(A, ld, 0, 0, m, n, d_transp, max_block, w, s));
  int out[2] = {-1, 0};
  ZLU_TRY(hipMemcpyAsync(out, w.first_zero, sizeof out, hipMemcpyDeviceToHost, s));
  ZLU_TRY(hipStreamSynchronize(s));
  *first_zero_pivot = out[0];
  *nb_transpositions = out[1];
  return hipSuccess;
}

}  // namespace

hipError_t getrf_device_complex(int kind, int64_t m, int64_t n, void* A, int64_t ld, int* d_transp, int64_t max_block,
                                int* first_zero_pivot, int* nb_transpositions, hipStream_t s) {
  if (max_block < 8) max_block = 8;
  if (kind == C64) return zgetrf_t<double>(m, n, static_cast<Z<double>*>(A), ld, d_transp, max_block, first_zero_pivot, nb_transpositions, s);
  if (kind == C32) return zgetrf_t<float>(m, n, static_cast<Z<float>*>(A), ld, d_transp, max_block, first_zero_pivot, nb_transpositions, s);
  return hipErrorInvalidValue;
}

}  // namespace eigamd
