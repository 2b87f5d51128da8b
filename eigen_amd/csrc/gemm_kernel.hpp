// MFMA GEMM kernels for gfx950 (MI355X): C = alpha * op(A) * op(B) + beta * C
// for float, double, complex<float>, complex<double>, column-major C.
//
// This replaces the reference's pack + micro-kernel pipeline
//   gemm_pack_lhs  (Eigen/src/Core/products/GeneralBlockPanelKernel.h:2090-2369)
//   gemm_pack_rhs  (GeneralBlockPanelKernel.h:2378-2603)
//   gebp_kernel    (GeneralBlockPanelKernel.h:1404-2073)
// driven by general_matrix_matrix_product::run (GeneralMatrixMatrix.h:52-207).
// On the GPU the "packing" is the global -> LDS tile copy (128-bit loads, no
// packed buffer in HBM) and the register micro-tile is a grid of MFMA tiles per
// wavefront.  The alpha scaling and the beta pre-pass of the BLAS wrapper
// (blas/level3_impl.h:62-66) are fused into the epilogue.
//
// Operand layouts.  Each operand is described by which logical dimension is
// contiguous in memory: "X-contiguous" (the non-reduction index m or n runs
// fastest: A for op(A)=A, B for op(B)=B^T) or "K-contiguous" (A^T / A^H, and
// plain B).  The LDS image keeps the same contiguous dimension as global
// memory, so every global read and LDS write is a 16-byte vector; only the
// fragment reads differ between the two layouts.
//
// MFMA operand roles are swapped with respect to the matrices: the MFMA "A"
// operand carries B-matrix columns and the MFMA "B" operand carries A-matrix
// rows, so the accumulator lanes run along m and the C stores of a wavefront
// are contiguous along C's columns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace eigamd {

enum Kind : int { F32 = 0, F64 = 1, C32 = 2, C64 = 3 };
// A second tile geometry per kind, used by the launcher for products of few large tiles (never a caller's kind).
constexpr int F64S = 4, F32S = 5, C64S = 6, C32S = 7;

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));

struct GemmParams {
  const void* A;
  const void* B;
  void* C;
  int64_t lda, ldb, ldc;  // leading dimensions, in elements
  int M, N, K;            // extents of this launch (relative to A, B, C)
  int tiles_m, tiles_n;   // grid decomposition
  double alpha[2];        // (re, im)
  double beta[2];
  int conj_a, conj_b;     // complex only: conjugate the operand's elements
  int group_m;            // tile rows walked together by consecutive blocks (L2 reuse)
  int xcd_remap;          // 1: give each XCD a contiguous run of the tile sequence
  int tri;                // 0 all of C, 1 only row >= col, 2 only row <= col (indices of the whole C:
  int tri_r0, tri_c0;     // this launch covers the region that starts at row tri_r0, column tri_c0)
  int ktri;               // structural zeros of a triangular operand: terms of the reduction known to
                          // vanish are skipped by whole k-tiles.  0 none; op(A)(row, k) = 0 for 1: k > row,
                          // 2: k < row; op(B)(k, col) = 0 for 3: k > col, 4: k < col
  int splits;             // > 1: the grid holds `splits` copies of the tile sequence; copy s reduces over
  int k_chunk;            // k in [s * k_chunk, min(K, (s + 1) * k_chunk)) (k_chunk a multiple of the k-tile)
  int64_t split_stride;   // and writes its product to C + s * split_stride (elements): partial sums that
                          // splitk_reduce_kernel adds up.  tri and ktri must be 0 then.
  int atomic_c;           // real kinds, beta == 1 only: C += alpha * acc by one floating-point atomic add per
                          // element, executed at the memory side, instead of load / add / store.  Each element
                          // receives exactly one add, so the result is the same bits on every run (and the same
                          // bits as the load / add / store form: one rounding of c + alpha * acc either way), but
                          // no wavefront ever waits for C: a short reduction (the k = 256 ... 1024 trailing updates
                          // of the LU) no longer pays the latency of a whole C tile per workgroup.
};

// Tile geometry per scalar kind.  WM x WN wavefronts per workgroup, each owning
// MT x NT MFMA tiles; BK is the k extent of one LDS stage (128 bytes of k for the
// K-contiguous operand).  The EIGAMD_* macros exist so that profiling builds can try
// other shapes (tools/build_variant.sh); the defaults are the shipped configuration.
// Shape = WM, WN, MT, NT, BK, workgroups per CU the register budget is sized for,
// and whether the fragments of k-step ks+1 are read explicitly ahead of the MFMAs of
// k-step ks (1) or the order is left to the compiler's scheduler (0).  Two optional values
// follow: the operand pipeline (1 = interleaved, see the main loop of gemm_kernel) and,
// for that pipeline, the divisor d: the LDS writes of the next tile are spread over the
// first KSTEPS / d k-steps and the global loads over as many k-steps after them.
#ifndef EIGAMD_F64_SHAPE
#define EIGAMD_F64_SHAPE 4, 2, 4, 4, 16, 1, 1, 1, 2
#endif
#ifndef EIGAMD_F32_SHAPE
#define EIGAMD_F32_SHAPE 2, 4, 4, 2, 32, 1, 1, 1, 2
#endif
// double, 128 x 64 tile, two workgroups per CU: a quarter of the large tile's work per workgroup, for products that
// would leave most compute units without a tile (or with a long single one).
#ifndef EIGAMD_F64S_SHAPE
#define EIGAMD_F64S_SHAPE 2, 2, 4, 2, 16, 2, 1, 1, 2
#endif
// float, 128 x 128 tile, two workgroups per CU (the large one is 256 x 256: 64 tiles for a 2048^2 result)
#ifndef EIGAMD_F32S_SHAPE
#define EIGAMD_F32S_SHAPE 2, 2, 2, 2, 32, 2, 1, 1, 2
#endif
// complex double 64 x 64 and complex float 64 x 128, two workgroups per CU
#ifndef EIGAMD_C64S_SHAPE
#define EIGAMD_C64S_SHAPE 2, 2, 2, 2, 16, 2, 1, 1, 2
#endif
#ifndef EIGAMD_C32S_SHAPE
#define EIGAMD_C32S_SHAPE 2, 2, 1, 2, 16, 2, 1, 1, 2
#endif
#ifndef EIGAMD_C64_SHAPE
#define EIGAMD_C64_SHAPE 2, 4, 4, 2, 16, 1, 1, 1, 2
#endif
#ifndef EIGAMD_C32_SHAPE
#define EIGAMD_C32_SHAPE 2, 4, 2, 2, 16, 1, 1, 1, 2
#endif

// Profiling builds can switch stages of the main loop off (results are then wrong; the
// timing shows what the remaining stages cost): 1 = no global loads, 2 = no LDS
// stores, 4 = no barrier.  Never set in the shipped library.
#ifndef EIGAMD_ABLATE
#define EIGAMD_ABLATE 0
#endif

template <int WM_, int WN_, int MT_, int NT_, int BK_, int PER_CU_, int PREFETCH_, int PIPE_ = 0, int SPREAD_DIV_ = 4>
struct Shape {
  static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_, BK = BK_, PER_CU = PER_CU_;
  static constexpr bool PREFETCH = PREFETCH_ != 0;
  static constexpr bool PIPE = PIPE_ != 0;
  static constexpr int SPREAD_DIV = SPREAD_DIV_;
};

template <int KIND> struct Traits;

template <> struct Traits<F32> {
  typedef float Elem;
  typedef float Real;
  typedef f16_t Acc;
  static constexpr bool kCplx = false;
  typedef Shape<EIGAMD_F32_SHAPE> S;
  static constexpr int TM = 32, KS = 2, NACC = 16;  // v_mfma_f32_32x32x2_f32
  static constexpr int WM = S::WM, WN = S::WN, MT = S::MT, NT = S::NT, BK = S::BK;
  static constexpr int PADX = 0, PADK = 1;
  static constexpr bool kVecWriteK = false;  // odd pitch: element-wise LDS writes
};
template <> struct Traits<F64> {
  typedef double Elem;
  typedef double Real;
  typedef d4_t Acc;
  static constexpr bool kCplx = false;
  typedef Shape<EIGAMD_F64_SHAPE> S;
  static constexpr int TM = 16, KS = 4, NACC = 4;  // v_mfma_f64_16x16x4_f64
  static constexpr int WM = S::WM, WN = S::WN, MT = S::MT, NT = S::NT, BK = S::BK;
  static constexpr int PADX = 16, PADK = 2;
  static constexpr bool kVecWriteK = true;
};
template <> struct Traits<F64S> : Traits<F64> {
  typedef Shape<EIGAMD_F64S_SHAPE> S;
  static constexpr int WM = S::WM, WN = S::WN, MT = S::MT, NT = S::NT, BK = S::BK;
};
template <> struct Traits<F32S> : Traits<F32> {
  typedef Shape<EIGAMD_F32S_SHAPE> S;
  static constexpr int WM = S::WM, WN = S::WN, MT = S::MT, NT = S::NT, BK = S::BK;
};
template <> struct Traits<C32> {
  typedef float2 Elem;
  typedef float Real;
  typedef f16_t Acc;
  static constexpr bool kCplx = true;
  typedef Shape<EIGAMD_C32_SHAPE> S;
  static constexpr int TM = 32, KS = 2, NACC = 16;
  static constexpr int WM = S::WM, WN = S::WN, MT = S::MT, NT = S::NT, BK = S::BK;
  static constexpr int PADX = 0, PADK = 1;
  static constexpr bool kVecWriteK = false;
};
template <> struct Traits<C64> {
  typedef double2 Elem;
  typedef double Real;
  typedef d4_t Acc;
  static constexpr bool kCplx = true;
  typedef Shape<EIGAMD_C64_SHAPE> S;
  static constexpr int TM = 16, KS = 4, NACC = 4;
  static constexpr int WM = S::WM, WN = S::WN, MT = S::MT, NT = S::NT, BK = S::BK;
  static constexpr int PADX = 0, PADK = 2;
  static constexpr bool kVecWriteK = true;
};

template <> struct Traits<C64S> : Traits<C64> {
  typedef Shape<EIGAMD_C64S_SHAPE> S;
  static constexpr int WM = S::WM, WN = S::WN, MT = S::MT, NT = S::NT, BK = S::BK;
};
template <> struct Traits<C32S> : Traits<C32> {
  typedef Shape<EIGAMD_C32S_SHAPE> S;
  static constexpr int WM = S::WM, WN = S::WN, MT = S::MT, NT = S::NT, BK = S::BK;
};

template <int KIND> struct Geometry {
  typedef Traits<KIND> T;
  static constexpr int BM = T::WM * T::MT * T::TM;
  static constexpr int BN = T::WN * T::NT * T::TM;
  static constexpr int THREADS = 64 * T::WM * T::WN;
  static constexpr int MIN_WAVES_PER_SIMD = T::S::PER_CU * THREADS / 256;
  static constexpr int BK = T::BK;
  static constexpr int ES = sizeof(typename T::Elem);
  static constexpr int EPC = 16 / ES;  // elements per 16-byte chunk
  static constexpr int lds_elems(bool kcontig, int bx) {
    return kcontig ? bx * (BK + T::PADK) : BK * (bx + T::PADX);
  }
};

__device__ __forceinline__ void mma(d4_t& c, double a, double b) {
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(f16_t& c, float a, float b) {
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float flip(float v, uint32_t mask) {
  return __uint_as_float(__float_as_uint(v) ^ mask);
}
__device__ __forceinline__ double flip(double v, uint32_t mask) {
  uint64_t u = (uint64_t)__double_as_longlong(v) ^ ((uint64_t)mask << 32);
  return __longlong_as_double((long long)u);
}

// Row (MFMA "A"-operand index, mapped to the n direction) of accumulator
// register r for this lane.  The column is lane % TM for every shape.
template <int KIND> __device__ __forceinline__ int acc_row(int r, int kk) {
  if constexpr (Traits<KIND>::TM == 16) return kk + 4 * r;       // f64 16x16x4
  else return (r & 3) + 8 * (r >> 2) + 4 * kk;                    // f32 32x32x2
}

// One operand tile: global -> registers -> LDS.
//   KC   : operand is K-contiguous in memory (and in LDS)
//   BX   : tile extent along the non-reduction dimension
// The tile is cut into 16-byte chunks; chunk c = tid + i * kThreads.
template <int KIND, bool KC, int BX> struct TileIO {
  typedef Traits<KIND> T;
  typedef Geometry<KIND> G;
  typedef typename T::Elem Elem;
  static constexpr int kThreads = G::THREADS;
  static constexpr int BK = T::BK;
  static constexpr int EPC = G::EPC;
  static constexpr int PITCH = KC ? (BK + T::PADK) : (BX + T::PADX);
  static constexpr int CPR = (KC ? BK : BX) / EPC;       // chunks per contiguous row
  static constexpr int NCHUNK = BX * BK / EPC / kThreads;  // chunks per thread
  static constexpr int ROWSTEP = kThreads / CPR;           // rows advanced per chunk index
  static_assert(BX * BK % (EPC * kThreads) == 0, "tile must split evenly over the block");
  static_assert(kThreads % CPR == 0, "a row must be covered by whole thread groups");

  int row0;   // first row (k for X-contiguous, x for K-contiguous) of this thread
  int col0;   // first element inside the row

  __device__ __forceinline__ void init(int tid) {
    row0 = tid / CPR;
    col0 = (tid % CPR) * EPC;
  }
  // Element offset of this thread's first chunk from the tile origin.
  __device__ __forceinline__ int64_t goffset(int64_t ld) const {
    return (int64_t)row0 * ld + col0;
  }
  // Unguarded: whole chunks, 16-byte aligned.
  __device__ __forceinline__ void load_one(uint4& r, int i, const Elem* p, int64_t ld) const {
    r = *reinterpret_cast<const uint4*>(p + (int64_t)(i * ROWSTEP) * ld);
  }
  __device__ __forceinline__ void load(uint4 (&r)[NCHUNK], const Elem* p, int64_t ld) const {
#pragma unroll
    for (int i = 0; i < NCHUNK; ++i) load_one(r[i], i, p, ld);
  }
  // Guarded: element-wise, zero outside [0,xrem) x [0,krem).  Needs only
  // element alignment.
  __device__ __forceinline__ void load_guard_one(uint4& r, int i, const Elem* p, int64_t ld,
                                                 int xrem, int krem) const {
    union { uint4 v; Elem e[EPC]; } u;
    const int row = row0 + i * ROWSTEP;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int col = col0 + e;
      const int x = KC ? row : col;
      const int k = KC ? col : row;
      Elem z = {};
      u.e[e] = (x < xrem && k < krem) ? p[(int64_t)(i * ROWSTEP) * ld + e] : z;
    }
    r = u.v;
  }
  __device__ __forceinline__ void load_guard(uint4 (&r)[NCHUNK], const Elem* p, int64_t ld,
                                             int xrem, int krem) const {
#pragma unroll
    for (int i = 0; i < NCHUNK; ++i) load_guard_one(r[i], i, p, ld, xrem, krem);
  }
  __device__ __forceinline__ void store_one(const uint4& r, int i, Elem* lds) const {
    Elem* dst = lds + (row0 + i * ROWSTEP) * PITCH + col0;
    if constexpr (!KC || T::kVecWriteK) {
      *reinterpret_cast<uint4*>(dst) = r;
    } else {
      union { uint4 v; Elem e[EPC]; } u;
      u.v = r;
#pragma unroll
      for (int e = 0; e < EPC; ++e) dst[e] = u.e[e];
    }
  }
  __device__ __forceinline__ void store(const uint4 (&r)[NCHUNK], Elem* lds) const {
#pragma unroll
    for (int i = 0; i < NCHUNK; ++i) store_one(r[i], i, lds);
  }
  // LDS element index of fragment element (x, k).
  __device__ static __forceinline__ int frag_index(int x, int k) {
    return KC ? x * PITCH + k : k * PITCH + x;
  }
};

template <int KIND, bool AK, bool BKC, bool GUARD>
__global__ void __launch_bounds__(Geometry<KIND>::THREADS, Geometry<KIND>::MIN_WAVES_PER_SIMD)
    gemm_kernel(const GemmParams p) {
  typedef Traits<KIND> T;
  typedef Geometry<KIND> G;
  typedef typename T::Elem Elem;
  typedef typename T::Real Real;
  typedef typename T::Acc Acc;
  constexpr int BM = G::BM, BN = G::BN, BK = G::BK;
  constexpr int TM = T::TM, KS = T::KS, MT = T::MT, NT = T::NT;
  constexpr int NPLANE = T::kCplx ? 2 : 1;
  typedef TileIO<KIND, AK, BM> IOA;
  typedef TileIO<KIND, BKC, BN> IOB;
  constexpr int LDS_A = G::lds_elems(AK, BM);
  constexpr int LDS_B = G::lds_elems(BKC, BN);

  __shared__ __attribute__((aligned(16))) Elem smem[2 * (LDS_A + LDS_B)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int xi = lane % TM;
  const int kk = lane / TM;
  const int wm0 = (wave % T::WM) * (MT * TM);
  const int wn0 = (wave / T::WM) * (NT * TM);

  // Block -> tile mapping.  Consecutive block ids are dealt round-robin over
  // the 8 XCDs; give each XCD a contiguous run of the tile sequence, and walk
  // the sequence in groups of tile rows so that co-resident blocks of one
  // XCD share A and B panels in that XCD's L2.
  int tile_m, tile_n;
  int split = 0;
  {
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    if (p.splits > 1) {
      split = bid / nblk;
      bid -= split * nblk;
    }
    const int q = nblk >> 3, rem = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int seq = p.xcd_remap ? (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx : bid;
    const int GM = p.group_m;
    const int per_group = GM * p.tiles_n;
    const int g = seq / per_group;
    const int first_m = g * GM;
    const int gsize = min(p.tiles_m - first_m, GM);
    const int in_g = seq - g * per_group;
    tile_m = first_m + in_g % gsize;
    tile_n = in_g / gsize;
  }
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  // Triangular destination: a tile that lies wholly outside the triangle has nothing to do.
  if (p.tri == 1 && p.tri_r0 + m0 + BM - 1 < p.tri_c0 + n0) return;
  if (p.tri == 2 && p.tri_r0 + m0 > p.tri_c0 + n0 + BN - 1) return;

  // Extent of this workgroup's reduction and where it starts.
  const int k_first = split * p.k_chunk;
  const int K = p.splits > 1 ? min(p.k_chunk, p.K - k_first) : p.K;
  const Elem* __restrict__ A = static_cast<const Elem*>(p.A) + (AK ? (int64_t)k_first : (int64_t)k_first * p.lda);
  const Elem* __restrict__ B = static_cast<const Elem*>(p.B) + (BKC ? (int64_t)k_first : (int64_t)k_first * p.ldb);

  IOA ioa; ioa.init(tid);
  IOB iob; iob.init(tid);
  // Tile origins: A tile = rows [m0, m0+BM) x k; B tile = k x cols [n0, n0+BN).
  const Elem* pa = A + (AK ? (int64_t)m0 * p.lda : (int64_t)m0) + ioa.goffset(p.lda);
  const Elem* pb = B + (BKC ? (int64_t)n0 * p.ldb : (int64_t)n0) + iob.goffset(p.ldb);
  const int64_t stepa = AK ? (int64_t)BK : (int64_t)BK * p.lda;
  const int64_t stepb = BKC ? (int64_t)BK : (int64_t)BK * p.ldb;
  const int mrem = p.M - m0, nrem = p.N - n0;

  uint4 ra[IOA::NCHUNK], rb[IOB::NCHUNK];

  Acc acc[NT][MT][NPLANE];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int c = 0; c < NPLANE; ++c)
#pragma unroll
        for (int r = 0; r < T::NACC; ++r) acc[j][i][c][r] = 0;

  // Sign masks for the complex product (see the accumulate step below).
  uint32_t mask_r = 0, mask_i = 0;
  if constexpr (T::kCplx) {
    const bool same = (p.conj_a != 0) == (p.conj_b != 0);
    mask_r = same ? 0x80000000u : 0u;
    mask_i = same ? 0u : 0x80000000u;
  }

  // k-tiles [t_begin, t_end) of this workgroup: all of them, or fewer when an operand is
  // triangular (the skipped tiles hold zeros only; tiles that straddle the diagonal are read,
  // their zeros are in memory).
  int t_begin = 0, t_end = (K + BK - 1) / BK;
  if (p.ktri == 1) t_end = min(t_end, (p.tri_r0 + m0 + BM - 1) / BK + 1);
  if (p.ktri == 2) t_begin = (p.tri_r0 + m0) / BK;
  if (p.ktri == 3) t_end = min(t_end, (p.tri_c0 + n0 + BN - 1) / BK + 1);
  if (p.ktri == 4) t_begin = (p.tri_c0 + n0) / BK;
  pa += (int64_t)t_begin * stepa;
  pb += (int64_t)t_begin * stepb;
  constexpr int KSTEPS = BK / KS;
  constexpr int NA = IOA::NCHUNK, NB = IOB::NCHUNK, NCH = NA + NB;
  constexpr bool PIPE = T::S::PIPE;
  // Software pipeline of the operand tiles.  The stage that tile t is multiplied from was
  // written one tile earlier; what differs is where the copy of the following tiles sits:
  //   PIPE = 0: tile t+1 is loaded from global memory before the MFMAs of tile t and
  //             written to LDS after them;
  //   PIPE = 1: the registers already hold tile t+1 (loaded during tile t-1); its LDS
  //             writes and then the global loads of tile t+2 are spread over the first
  //             k-steps of tile t, between the MFMAs, so no wavefront has a store-only
  //             phase in front of the barrier.
  // Work is placed in "slots": slot ks is issued in front of k-step ks, slot KSTEPS after
  // the last one.
  constexpr int LEAD = PIPE ? 2 : 1;
  constexpr int SPREAD = KSTEPS / T::S::SPREAD_DIV > 0 ? KSTEPS / T::S::SPREAD_DIV : 1;
  auto store_slot = [](int c) constexpr { return PIPE ? c * SPREAD / NCH : KSTEPS; };
  auto load_slot = [](int c) constexpr {
    const int s = SPREAD + c * SPREAD / NCH;
    return PIPE ? (s < KSTEPS ? s : KSTEPS - 1) : 0;
  };

  auto load_chunk = [&](int c, bool guard, int krem) {
    if (c < NA) {
      if (guard) ioa.load_guard_one(ra[c < NA ? c : 0], c, pa, p.lda, mrem, krem);
      else ioa.load_one(ra[c < NA ? c : 0], c, pa, p.lda);
    } else {
      const int d = c < NA ? 0 : c - NA;
      if (guard) iob.load_guard_one(rb[d], d, pb, p.ldb, nrem, krem);
      else iob.load_one(rb[d], d, pb, p.ldb);
    }
  };
  auto store_chunk = [&](int c, Elem* stage) {
    if (c < NA) ioa.store_one(ra[c < NA ? c : 0], c, stage);
    else iob.store_one(rb[c < NA ? 0 : c - NA], c < NA ? 0 : c - NA, stage + LDS_A);
  };
  // Whole tile `t` into the staging registers (prologue only).
  auto load_tile = [&](int t) {
    const int krem = K - t * BK;
    const bool guard = GUARD || krem < BK;
#pragma unroll
    for (int c = 0; c < NCH; ++c) load_chunk(c, guard, krem);
    pa += stepa;
    pb += stepb;
  };

  if (t_begin < t_end) {
    load_tile(t_begin);
#pragma unroll
    for (int c = 0; c < NCH; ++c) store_chunk(c, smem + (t_begin & 1) * (LDS_A + LDS_B));
    if (PIPE && t_begin + 1 < t_end) load_tile(t_begin + 1);
  }
  __syncthreads();

  // Per-lane fragment origins inside a stage.
  const int fa0 = IOA::frag_index(wm0 + xi, kk);
  const int fb0 = IOB::frag_index(wn0 + xi, kk);

  // Epilogue state, set up in front of the main loop because the first loads of C are issued
  // from inside it.  C = alpha * acc + beta * C; beta == 0 never reads C
  // (blas/level3_impl.h:64: NaN/Inf in C must not propagate).
  const bool use_c = (p.beta[0] != 0.0 || p.beta[1] != 0.0) && !p.atomic_c;
  // Triangular destinations (the symmetric rank-k update): elements outside the triangle are
  // neither read nor written; only the epilogue knows about it.
  auto in_tri = [&](int row, int col) {
    return p.tri == 0 || (p.tri == 1 ? p.tri_r0 + row >= p.tri_c0 + col : p.tri_r0 + row <= p.tri_c0 + col);
  };
  const int mrow = wm0 + xi;                       // + i * TM
  const int nrow = wn0 + acc_row<KIND>(0, kk);     // + j * TM + acc_row(r, 0)
  Elem* __restrict__ cbase = static_cast<Elem*>(p.C) + split * p.split_stride + (int64_t)(m0 + mrow) + (int64_t)(n0 + nrow) * p.ldc;
  // C is read one column group (the NT index) at a time into a ring of kEpiStages register
  // buffers.  Where a group is small (32 VGPRs: f64) the ring has three buffers and the loads
  // of group 0 are issued at the top of the LAST k-tile, whose operand staging registers are
  // already free, so that they arrive under its MFMAs; groups 1 and 2 follow right after the
  // loop and group 3 when buffer 0 has been stored.  (k = 256, 32768^2: 54.2 -> 55.5 TFLOP/s.)
  constexpr int kGroupRegs = MT * T::NACC * (int)sizeof(Elem) / 4;
  constexpr int kEpiStages = kGroupRegs <= 32 ? (NT >= 3 ? 3 : 2) : 1;
  constexpr bool kEarlyC = kEpiStages >= 2;
  Elem cv[kEpiStages][MT][T::NACC];
  auto load_group = [&](int j, Elem (&dst)[MT][T::NACC]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < T::NACC; ++r) {
        const int nn = j * TM + acc_row<KIND>(r, 0);
        Elem z = {};
        dst[i][r] = ((!GUARD || (mrow + i * TM < mrem && nrow + nn < nrem)) && in_tri(m0 + mrow + i * TM, n0 + nrow + nn))
                        ? cbase[i * TM + (int64_t)nn * p.ldc] : z;
      }
  };
  bool c0_loaded = false;

  // One tile of the reduction.  STEADY: tiles t+1 and t+LEAD exist and are whole, nothing
  // is tested; otherwise both are tested and partial tiles take the guarded loads.
  auto tile = [&](int t, auto steady_c) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const Elem* sa = smem + (t & 1) * (LDS_A + LDS_B) + fa0;
    const Elem* sb = smem + (t & 1) * (LDS_A + LDS_B) + LDS_A + fb0;
    Elem* next_stage = smem + ((t + 1) & 1) * (LDS_A + LDS_B);
    const int krem = K - (t + LEAD) * BK;
    const bool do_store = (STEADY || t + 1 < t_end) && !(EIGAMD_ABLATE & 2);
    const bool do_load = (STEADY || t + LEAD < t_end) && !(EIGAMD_ABLATE & 1);
    // (a bounds-checked launch keeps its row / column checks in the steady loop; only the k check is dropped there)
    const bool guard = STEADY ? GUARD : (GUARD || krem < BK);
    // In the interleaved pipeline the position of the copies among the MFMAs is the
    // point, so the instruction scheduler may not move anything across a used slot.
    auto slot = [&](int s) {
      bool used = false;
#pragma unroll
      for (int c = 0; c < NCH; ++c) used = used || store_slot(c) == s || load_slot(c) == s;
      if (PIPE && used) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (store_slot(c) == s && do_store) store_chunk(c, next_stage);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (load_slot(c) == s && do_load) load_chunk(c, guard, krem);
      if (PIPE && used) __builtin_amdgcn_sched_barrier(0);
    };

    // With PREFETCH the fragments of k-step ks+1 are read while the MFMAs of k-step ks
    // issue; without it each k-step reads its own fragments and the compiler interleaves.
    if constexpr (!STEADY && kEarlyC) {
      if (use_c && t + 1 == t_end) {
        load_group(0, cv[0]);
        c0_loaded = true;
      }
    }
    Elem fa[2][MT], fb[2][NT];
    if constexpr (T::S::PREFETCH) {
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[0][i] = sa[IOA::frag_index(i * TM, 0)];
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[0][j] = sb[IOB::frag_index(j * TM, 0)];
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      slot(ks);
      const int cur = T::S::PREFETCH ? (ks & 1) : 0, nxt = cur ^ 1;
      if constexpr (T::S::PREFETCH) {
        if (ks + 1 < KSTEPS) {
#pragma unroll
          for (int i = 0; i < MT; ++i) fa[nxt][i] = sa[IOA::frag_index(i * TM, (ks + 1) * KS)];
#pragma unroll
          for (int j = 0; j < NT; ++j) fb[nxt][j] = sb[IOB::frag_index(j * TM, (ks + 1) * KS)];
        }
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = sa[IOA::frag_index(i * TM, ks * KS)];
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[0][j] = sb[IOB::frag_index(j * TM, ks * KS)];
      }
      if constexpr (!T::kCplx) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) mma(acc[j][i][0], fb[cur][j], fa[cur][i]);
      } else {
        // With sa, sb = -1 for a conjugated operand and +1 otherwise:
        //   Re += br*ar - (sa*sb) * bi*ai
        //   Im  = sa * ( br*ai + (sa*sb) * bi*ar )   (the leading sa is applied in the epilogue)
        // so one sign flip of bi per accumulator covers all four cases.
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const Real br = fb[cur][j].x, bi = fb[cur][j].y;
          const Real bi_r = flip(bi, mask_r);
          const Real bi_i = flip(bi, mask_i);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            mma(acc[j][i][0], br, fa[cur][i].x);
            mma(acc[j][i][1], br, fa[cur][i].y);
            mma(acc[j][i][0], bi_r, fa[cur][i].y);
            mma(acc[j][i][1], bi_i, fa[cur][i].x);
          }
        }
      }
    }
    slot(KSTEPS);
    if (do_load) {
      pa += stepa;
      pb += stepb;
    }
    if (!(EIGAMD_ABLATE & 4)) __syncthreads();
  };

  int t = t_begin;
  {
    const int nfull = min(K / BK, t_end);
    for (; t + LEAD < nfull; ++t) tile(t, std::true_type{});
  }
  for (; t < t_end; ++t) tile(t, std::false_type{});

  const Real alr = (Real)p.alpha[0], ali = (Real)p.alpha[1];
  const Real ber = (Real)p.beta[0], bei = (Real)p.beta[1];
  const Real sgn = p.conj_a ? (Real)-1 : (Real)1;
  if constexpr (!T::kCplx) {
    if (p.atomic_c) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < T::NACC; ++r) {
            const int nn = j * TM + acc_row<KIND>(r, 0);
            if ((!GUARD || (mrow + i * TM < mrem && nrow + nn < nrem)) && in_tri(m0 + mrow + i * TM, n0 + nrow + nn))
              unsafeAtomicAdd(&cbase[i * TM + (int64_t)nn * p.ldc], alr * acc[j][i][0][r]);
          }
      return;
    }
  }
  if (use_c) {
    if (!c0_loaded) load_group(0, cv[0]);
#pragma unroll
    for (int j = 1; j < kEpiStages && j < NT; ++j) load_group(j, cv[j]);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < T::NACC; ++r) {
        const int nn = j * TM + acc_row<KIND>(r, 0);
        const Elem cin = cv[j % kEpiStages][i][r];
        Elem v;
        if constexpr (!T::kCplx) {
          v = alr * acc[j][i][0][r];
          if (use_c) v = fma(ber, cin, v);
        } else {
          const Real pr = acc[j][i][0][r], pi = sgn * acc[j][i][1][r];
          v.x = alr * pr - ali * pi;
          v.y = alr * pi + ali * pr;
          if (use_c) {
            v.x += ber * cin.x - bei * cin.y;
            v.y += ber * cin.y + bei * cin.x;
          }
        }
        if ((!GUARD || (mrow + i * TM < mrem && nrow + nn < nrem)) && in_tri(m0 + mrow + i * TM, n0 + nrow + nn))
          cbase[i * TM + (int64_t)nn * p.ldc] = v;
      }
    if (use_c && j + kEpiStages < NT) load_group(j + kEpiStages, cv[j % kEpiStages]);
  }
}

// C = beta * C over an m x n window (the k == 0 case of the BLAS wrapper,
// blas/level3_impl.h:62-69).
template <int KIND>
__global__ void scale_kernel(void* Cv, int64_t ldc, int M, int N, double beta_re, double beta_im, int tri) {
  typedef typename Traits<KIND>::Elem Elem;
  typedef typename Traits<KIND>::Real Real;
  Elem* C = static_cast<Elem*>(Cv);
  const int64_t total = (int64_t)M * N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = idx / M, m = idx - n * M;
    if (tri != 0 && (tri == 1 ? m < n : m > n)) continue;
    Elem* dst = C + m + n * ldc;
    if constexpr (!Traits<KIND>::kCplx) {
      *dst = beta_re == 0.0 ? (Real)0 : (Real)beta_re * *dst;
    } else {
      Elem v = {};
      if (beta_re != 0.0 || beta_im != 0.0) {
        const Elem c = *dst;
        v.x = (Real)beta_re * c.x - (Real)beta_im * c.y;
        v.y = (Real)beta_re * c.y + (Real)beta_im * c.x;
      }
      *dst = v;
    }
  }
}

// C = alpha * (P_0 + P_1 + ... + P_{S-1}) + beta * C, the partial products of a reduction split
// over S copies of the tile sequence (GemmParams::splits); P_s is the dense m x n matrix at
// P + s * m * n.  The sum runs in the order of s whatever order the copies finished in, so a
// product gives the same bits on every run.  beta == 0 never reads C.
template <int KIND>
__global__ void splitk_reduce_kernel(const void* Pv, int splits, void* Cv, int64_t ldc, int M, int N,
                                     double alpha_re, double alpha_im, double beta_re, double beta_im) {
  typedef typename Traits<KIND>::Elem Elem;
  typedef typename Traits<KIND>::Real Real;
  const Elem* P = static_cast<const Elem*>(Pv);
  Elem* C = static_cast<Elem*>(Cv);
  const int64_t total = (int64_t)M * N;
  const bool use_c = beta_re != 0.0 || beta_im != 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = idx / M, m = idx - n * M;
    Elem sum = P[idx];
    for (int s = 1; s < splits; ++s) {
      const Elem v = P[idx + s * total];
      if constexpr (!Traits<KIND>::kCplx) sum += v;
      else { sum.x += v.x; sum.y += v.y; }
    }
    Elem* dst = C + m + n * ldc;
    if constexpr (!Traits<KIND>::kCplx) {
      Real v = (Real)alpha_re * sum;
      if (use_c) v = fma((Real)beta_re, *dst, v);
      *dst = v;
    } else {
      const Real alr = (Real)alpha_re, ali = (Real)alpha_im;
      Elem v;
      v.x = alr * sum.x - ali * sum.y;
      v.y = alr * sum.y + ali * sum.x;
      if (use_c) {
        const Elem c = *dst;
        v.x += (Real)beta_re * c.x - (Real)beta_im * c.y;
        v.y += (Real)beta_re * c.y + (Real)beta_im * c.x;
      }
      *dst = v;
    }
  }
}

}  // namespace eigamd
