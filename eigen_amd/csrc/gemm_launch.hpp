// Host-side launcher for the MFMA GEMM kernels: splits C into an interior
// region of whole block tiles (vector loads, no bounds checks) and up to two
// guarded edge strips.  One translation unit per scalar kind instantiates
// launch_gemm<KIND>.
#pragma once
#include <algorithm>
#include <cstdlib>

#include "gemm_kernel.hpp"

namespace eigamd {

// op codes shared by every entry point
enum Op : int { OP_N = 0, OP_T = 1, OP_C = 2, OP_CONJ = 3 /* conj, no transpose */ };

struct GemmCall {
  int kind;
  int opa, opb;
  int64_t m, n, k;
  double alpha[2], beta[2];
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  bool serial_edges = false;  // keep the edge launches on the caller's stream (callers that partition the CUs by stream)
  int tri = 0;  // 1 / 2: write (and read) only the lower / upper triangle of C (row >= col / row <= col)
  int ktri = 0; // structural zeros of a triangular operand (GemmParams::ktri)
  // Caller-provided memory for the partial products of a split reduction (launch_whole) in
  // place of splitk_workspace(): the launch path then allocates nothing.
  void* partials = nullptr;
  size_t partials_bytes = 0;
  bool partials_keep_helpers = false;  // with `partials`: the edge strips may still run on the library's helper streams
  // Real kinds, beta == 1: accumulate into C with memory-side atomic adds (GemmParams::atomic_c).  C must
  // be ordinary device memory (hipMalloc); ignored for the complex kinds and for other beta.
  bool atomic_c = false;
  // beta == 0 only.  When the launcher cuts the whole reduction into pieces, the pass that adds the pieces up writes
  // alpha * sum to split_dest (leading dimension split_ldd) instead of C and sets *split_dest_used; split_dest may be
  // one of the operands -- every read of A and B is over before that pass starts.  Otherwise the product goes to C as
  // usual and the flag stays as it was.  (The triangular solves' leaves: B_j <- inv_j * B_j without a copy.)
  void* split_dest = nullptr;
  int64_t split_ldd = 0;
  bool* split_dest_used = nullptr;
};

// Block-to-tile mapping knobs; the defaults are what the shipped library uses, the
// environment overrides exist for profiling experiments only.
struct Tuning {
  int group_m = 0;  // 0: the kind's default (default_group_m)
  int xcd_remap = 1;
  int extra_lds = 0;  // bytes of unused dynamic LDS per block (lowers blocks per CU)
};
// Tile rows walked together by consecutive workgroups.  The 32 workgroups an XCD runs at a time
// should cover a square patch of C, which needs the fewest distinct rows of A and columns of B
// per k-step in that XCD's L2: 4 x 8 tiles of 256 x 128 for f64 (L2 hit 80 % against 65 % with
// groups of 8, L2 <-> fabric reads per 16384^3 product 80 GB against 148 GB), 8 x 4 for the others.
template <int KIND> constexpr int default_group_m() { return KIND == F64 ? 4 : 8; }
// Workgroups of a kind's kernel that share a compute unit.
template <int KIND> constexpr int workgroups_per_cu() { return Traits<KIND>::S::PER_CU; }

inline const Tuning& tuning() {
  static const Tuning t = [] {
    Tuning v;
    if (const char* e = std::getenv("EIGAMD_GROUP_M")) v.group_m = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("EIGAMD_XCD_REMAP")) v.xcd_remap = std::atoi(e) != 0;
    if (const char* e = std::getenv("EIGAMD_EXTRA_LDS")) v.extra_lds = std::max(0, std::atoi(e));
    return v;
  }();
  return t;
}

// One instance per device: scratch memory, streams and events belong to the device that was
// current when they were created.
constexpr int kMaxDevices = 16;
inline int current_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
  return d;
}

// Two helper streams per device for the bounds-checked edge launches of a product whose extents
// are not multiples of the tile.  Queued behind the interior launch on one stream, each edge strip
// (a handful of workgroups, each with the whole reduction in front of it) would run alone on the
// chip for as long as the interior did: a 4000^3 dgemm took 4.2 ms against 2.2 ms at 4096^3.  On
// their own streams they fill the compute units the interior's last round leaves free.
// fork(): the helper streams wait for everything queued on `s`; join(): `s` waits for them.
// Defined in blas_abi.hip; both return hipSuccess and leave *a = *b = s when forking is off.
hipError_t edge_streams_fork(hipStream_t s, hipStream_t* a, hipStream_t* b);
hipError_t edge_streams_join(hipStream_t s, hipStream_t a, hipStream_t b);

// Device memory for the partial products of a reduction split over several workgroups per tile
// (launch_whole): at least `bytes`, owned by the library, one of a small pool per device.  A buffer
// goes to the stream that used it last when there is one (work queued on a stream runs in order);
// otherwise the least recently used buffer is taken and `s` is made to wait for its last user.
// *slot identifies the buffer for splitk_release, which the caller invokes once the work that reads
// the buffer has been queued on `s`.  nullptr: none to be had (switched off by EIGAMD_SPLITK=0,
// hipStreamPerThread -- one handle, a different stream in every host thread --, out of memory); the
// product then runs unsplit.  Defined in blas_abi.hip.
void* splitk_acquire(hipStream_t s, size_t bytes, int* slot);
void splitk_release(int slot, hipStream_t s);

hipError_t launch_gemm_f32(const GemmCall&, hipStream_t);
hipError_t launch_gemm_f64(const GemmCall&, hipStream_t);
hipError_t launch_gemm_c32(const GemmCall&, hipStream_t);
hipError_t launch_gemm_c64(const GemmCall&, hipStream_t);
hipError_t launch_gemm_f64_small(const GemmCall&, hipStream_t);  // the 128 x 64 tile (gemm_inst_f64s.hip)
hipError_t launch_gemm_f32_small(const GemmCall&, hipStream_t);  // the 128 x 128 tile (gemm_inst_f32s.hip)
hipError_t launch_gemm_c64_small(const GemmCall&, hipStream_t);  // 64 x 64 (gemm_inst_c64s.hip)
hipError_t launch_gemm_c32_small(const GemmCall&, hipStream_t);  // 64 x 128 (gemm_inst_c32s.hip)

template <int KIND, bool AK, bool BKC, bool GUARD>
static hipError_t launch_region(const GemmCall& c, int64_t r0, int64_t c0, int64_t rows,
                                int64_t cols, hipStream_t stream, int splits = 0, int k_chunk = 0,
                                void* partials = nullptr) {
  typedef Geometry<KIND> G;
  typedef typename Traits<KIND>::Elem Elem;
  if (rows <= 0 || cols <= 0) return hipSuccess;
  GemmParams p;
  const Elem* A = static_cast<const Elem*>(c.A);
  const Elem* B = static_cast<const Elem*>(c.B);
  Elem* C = static_cast<Elem*>(c.C);
  p.A = A + (AK ? r0 * c.lda : r0);
  p.B = B + (BKC ? c0 * c.ldb : c0);
  p.C = C + r0 + c0 * c.ldc;
  p.lda = c.lda; p.ldb = c.ldb; p.ldc = c.ldc;
  p.M = (int)rows; p.N = (int)cols; p.K = (int)c.k;
  p.tiles_m = (int)((rows + G::BM - 1) / G::BM);
  p.tiles_n = (int)((cols + G::BN - 1) / G::BN);
  p.alpha[0] = c.alpha[0]; p.alpha[1] = c.alpha[1];
  p.beta[0] = c.beta[0]; p.beta[1] = c.beta[1];
  p.conj_a = (c.opa == OP_C || c.opa == OP_CONJ);
  p.conj_b = (c.opb == OP_C || c.opb == OP_CONJ);
  p.group_m = tuning().group_m > 0 ? tuning().group_m : default_group_m<KIND>();
  // A triangular destination leaves about half of the tiles empty, a triangular operand gives
  // the tiles very different reduction lengths: contiguous runs of the tile sequence per XCD
  // would give the XCDs very different amounts of work, round-robin does not.
  p.xcd_remap = (c.tri || c.ktri) ? 0 : tuning().xcd_remap;
  p.tri = c.tri; p.tri_r0 = (int)r0; p.tri_c0 = (int)c0;
  p.ktri = c.ktri;
  p.splits = splits; p.k_chunk = k_chunk; p.split_stride = rows * cols;
  p.atomic_c = (c.atomic_c && !Traits<KIND>::kCplx && c.beta[0] == 1.0 && c.beta[1] == 0.0 && splits <= 1) ? 1 : 0;
  if (splits > 1) {  // plain partial products, dense, one after the other
    p.C = partials; p.ldc = rows;
    p.alpha[0] = 1.0; p.alpha[1] = 0.0; p.beta[0] = 0.0; p.beta[1] = 0.0;
  }
  const int64_t nblk = (int64_t)p.tiles_m * p.tiles_n * (splits > 1 ? splits : 1);
  if (nblk > 0x7fffffff) return hipErrorInvalidValue;
  hipLaunchKernelGGL((gemm_kernel<KIND, AK, BKC, GUARD>), dim3((unsigned)nblk), dim3(G::THREADS),
                     tuning().extra_lds, stream, p);
  return hipGetLastError();
}

// What one workgroup's tile costs, in microseconds: t0 + tk * k (prologue, epilogue and launch
// share; one k of the main loop).  Measured on device-resident square products of 512 and 1024,
// each a single round of tiles (f64 0.136 / 0.259 ms, f32 0.149 / 0.269 ms, c64 and c32 both
// 0.255 / 0.494 ms).
template <int KIND> struct TileCost;
// The pass that adds partial products up reads them soon after they were written (they are still
// in the last-level cache): 4 TB/s.  With 2.5 TB/s and a 15 % margin the model left the 128-row
// products of a ztrsm unsplit; split, the 4096^2 solve takes 6.7 ms against 7.6.
constexpr double kReduceBytesPerUs = 4.0e6;
template <> struct TileCost<F64> { static constexpr double t0 = 13.0, tk = 0.240; };
template <> struct TileCost<F32> { static constexpr double t0 = 29.0, tk = 0.234; };
template <> struct TileCost<C64> { static constexpr double t0 = 13.0, tk = 0.467; };
template <> struct TileCost<C32> { static constexpr double t0 = 16.0, tk = 0.467; };
// (per workgroup with two on a compute unit: 128 x 64 x k at about 0.8 of half a CU's rate)
template <> struct TileCost<F64S> { static constexpr double t0 = 9.0, tk = 0.125; };
template <> struct TileCost<F32S> { static constexpr double t0 = 12.0, tk = 0.125; };
template <> struct TileCost<C64S> { static constexpr double t0 = 9.0, tk = 0.248; };
template <> struct TileCost<C32S> { static constexpr double t0 = 10.0, tk = 0.248; };

// A product of few tiles leaves most of the 256 compute units idle while each busy one walks the
// whole reduction (dgemm 1024^3: 32 tiles, 0.26 ms, 8 TFLOP/s).  Returns how many pieces to cut
// the reduction into (1: leave it whole) and the piece length, from a model of both variants: rounds of
// 256 workgroups times the tile cost, plus for the split form the pass that adds the partial
// products up (pieces + 2 matrices at kReduceBytesPerUs, and a launch).
template <int KIND>
static int choose_splits(const GemmCall& c, int* k_chunk, double* cost_out = nullptr) {
  typedef Geometry<KIND> G;
  typedef TileCost<KIND> TC;
  constexpr int kCUs = 256 * workgroups_per_cu<KIND>(), kMaxSplits = 16;  // (workgroups that run side by side)
  constexpr size_t kMaxBytes = (size_t)256 << 20;
  *k_chunk = (int)c.k;
  if (c.tri || c.ktri || c.serial_edges || c.k < 8 * G::BK) return 1;
  const int64_t tiles = ((c.m + G::BM - 1) / G::BM) * ((c.n + G::BN - 1) / G::BN);
  if (tiles >= 2 * kCUs) return 1;
  auto rounds = [&](int64_t blocks) { return (double)((blocks + kCUs - 1) / kCUs); };
  const double mn_bytes = (double)c.m * c.n * G::ES;
  const double whole = rounds(tiles) * (TC::t0 + TC::tk * c.k);
  double best = whole * 0.9;  // a split has to win clearly
  int best_s = 1;
  for (int s = 2; s <= kMaxSplits; ++s) {
    const int64_t kc = ((c.k + s - 1) / s + G::BK - 1) / G::BK * G::BK;
    if (kc < 4 * G::BK) break;
    const int pieces = (int)((c.k + kc - 1) / kc);
    if (pieces != s) continue;  // the same pieces as a smaller s gave
    if (mn_bytes * s > (double)kMaxBytes) break;
    const double cost = rounds(tiles * s) * (TC::t0 + TC::tk * kc) + 6.0 + mn_bytes * (s + 2) / kReduceBytesPerUs;
    if (cost < best) { best = cost; best_s = s; *k_chunk = (int)kc; }
  }
  if (cost_out) *cost_out = best_s > 1 ? best : whole;
  return best_s;
}

// A product of a few rounds of tiles whose last round is mostly empty (dgemm 5120^3: 800 tiles,
// 3.1 rounds, the time of 4) is cut in two along tile rows or tile columns: a first part of whole
// rounds, and the rest, whose reduction is split so that its workgroups are short.  Returns false
// when that does not pay.  `cut`: first row (by_rows) or column of the second part.
struct TailPlan {
  bool by_rows;
  int64_t cut;
  int splits, k_chunk;
};
template <int KIND>
static bool choose_tail(const GemmCall& c, TailPlan* plan, double* cost_out = nullptr) {
  typedef Geometry<KIND> G;
  typedef TileCost<KIND> TC;
  constexpr int kCUs = 256 * workgroups_per_cu<KIND>(), kMaxSplits = 16, kMaxRounds = 12;
  constexpr size_t kMaxBytes = (size_t)256 << 20;
  if (c.tri || c.ktri || c.serial_edges || c.k < 8 * G::BK) return false;
  static const bool off = [] { const char* e = std::getenv("EIGAMD_TAILSPLIT"); return e && std::atoi(e) == 0; }();
  if (off) return false;  // (experiments)
  const int64_t tm = (c.m + G::BM - 1) / G::BM, tn = (c.n + G::BN - 1) / G::BN;
  const int64_t tiles = tm * tn;
  const int64_t rounds = (tiles + kCUs - 1) / kCUs;
  if (rounds < 2 || rounds > kMaxRounds || tiles % kCUs == 0) return false;
  const double t_tile = TC::t0 + TC::tk * c.k;
  const int64_t over = tiles - (rounds - 1) * kCUs;  // tiles of the last round
  double best = 0.92 * rounds * t_tile;
  bool found = false;
  for (int by_rows = 0; by_rows < 2; ++by_rows) {
    const int64_t across = by_rows ? tn : tm;            // tiles per tile row / tile column
    const int64_t lines = (over + across - 1) / across;  // tile rows / columns that go to the second part
    if (lines >= (by_rows ? tm : tn)) continue;
    const int64_t tail_tiles = lines * across;
    const int64_t cut = by_rows ? (tm - lines) * G::BM : (tn - lines) * G::BN;
    const double tail_elems = by_rows ? (double)(c.m - cut) * c.n : (double)c.m * (c.n - cut);
    for (int s = 1; s <= kMaxSplits; ++s) {
      const int64_t kc = ((c.k + s - 1) / s + G::BK - 1) / G::BK * G::BK;
      if (s > 1 && kc < 4 * G::BK) break;
      if ((int)((c.k + kc - 1) / kc) != s) continue;
      if (s > 1 && tail_elems * G::ES * s > (double)kMaxBytes) break;
      const double tail_rounds = (double)((tail_tiles * s + kCUs - 1) / kCUs);
      const double cost = (rounds - 1) * t_tile + tail_rounds * (TC::t0 + TC::tk * kc) + 4.0 +
                          (s > 1 ? 6.0 + tail_elems * G::ES * (s + 2) / kReduceBytesPerUs : 0.0);
      if (cost < best) {
        best = cost; found = true;
        plan->by_rows = by_rows != 0; plan->cut = cut; plan->splits = s; plan->k_chunk = (int)kc;
      }
    }
  }
  if (cost_out) *cost_out = best;
  return found;
}

// alpha * (sum of the partial products) + beta * C over a region of C
template <int KIND>
static hipError_t reduce_partials(const GemmCall& c, int64_t r0, int64_t c0, int64_t rows, int64_t cols, int splits,
                                  const void* partials, hipStream_t stream) {
  typedef typename Traits<KIND>::Elem Elem;
  const int64_t total = rows * cols;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL((splitk_reduce_kernel<KIND>), dim3(blocks), dim3(256), 0, stream, partials, splits,
                     static_cast<Elem*>(c.C) + r0 + c0 * c.ldc, c.ldc, (int)rows, (int)cols, c.alpha[0], c.alpha[1],
                     c.beta[0], c.beta[1]);
  return hipGetLastError();
}

// One launch over all of C, its reduction split where that pays; or two, when the last round
// of tiles would be mostly empty.
template <int KIND, bool AK, bool BKC, bool GUARD>
static hipError_t launch_whole(const GemmCall& c, hipStream_t stream) {
  typedef Geometry<KIND> G;
  // Three forms, priced by the same model: one launch; one launch with the reduction cut into equal
  // pieces for every tile; whole rounds of tiles as they are plus the leftover tile rows / columns with
  // their reduction cut (dgemm 3072^3, 288 tiles: three equal pieces took 1.19 ms, 240 whole tiles plus
  // 48 tiles in five pieces take less).
  int k_chunk = 0;
  double split_cost = 0.0, tail_cost = 0.0;
  TailPlan t;
  int splits = choose_splits<KIND>(c, &k_chunk, &split_cost);
  const bool tail = choose_tail<KIND>(c, &t, &tail_cost);
  if (splits > 1 && tail && tail_cost < split_cost) splits = 1;
  if (splits > 1) {
    const size_t bytes = (size_t)splits * c.m * c.n * G::ES;
    int slot = -1;
    if (void* partials = c.partials ? (bytes <= c.partials_bytes ? c.partials : nullptr) : splitk_acquire(stream, bytes, &slot)) {
      hipError_t e = launch_region<KIND, AK, BKC, GUARD>(c, 0, 0, c.m, c.n, stream, splits, k_chunk, partials);
      const bool redirect = c.split_dest != nullptr && c.beta[0] == 0.0 && c.beta[1] == 0.0;
      GemmCall cr = c;
      if (redirect) { cr.C = c.split_dest; cr.ldc = c.split_ldd; }
      if (e == hipSuccess) e = reduce_partials<KIND>(cr, 0, 0, c.m, c.n, splits, partials, stream);
      if (e == hipSuccess && redirect && c.split_dest_used) *c.split_dest_used = true;
      if (slot >= 0) splitk_release(slot, stream);
      return e;
    }
    return launch_region<KIND, AK, BKC, GUARD>(c, 0, 0, c.m, c.n, stream);
  }
  if (tail) {
    const int64_t r0 = t.by_rows ? t.cut : 0, c0 = t.by_rows ? 0 : t.cut;
    const int64_t rows = c.m - r0, cols = c.n - c0;
    const size_t bytes = (size_t)t.splits * rows * cols * G::ES;
    int slot = -1;
    void* partials = t.splits <= 1 ? nullptr
                     : c.partials  ? (bytes <= c.partials_bytes ? c.partials : nullptr)
                                   : splitk_acquire(stream, bytes, &slot);
    if (t.splits == 1 || partials) {
      hipError_t e = launch_region<KIND, AK, BKC, GUARD>(c, 0, 0, t.by_rows ? t.cut : c.m, t.by_rows ? c.n : t.cut, stream);
      if (e == hipSuccess && t.splits == 1) return launch_region<KIND, AK, BKC, GUARD>(c, r0, c0, rows, cols, stream);
      if (e == hipSuccess) e = launch_region<KIND, AK, BKC, GUARD>(c, r0, c0, rows, cols, stream, t.splits, t.k_chunk, partials);
      if (e == hipSuccess) e = reduce_partials<KIND>(c, r0, c0, rows, cols, t.splits, partials, stream);
      if (slot >= 0) splitk_release(slot, stream);
      return e;
    }
  }
  return launch_region<KIND, AK, BKC, GUARD>(c, 0, 0, c.m, c.n, stream);
}

// Modelled time (microseconds) of the form launch_whole would choose for this product with this kind's tile.
// Where two workgroups share a compute unit a round that is not full does not cost a whole round: a workgroup
// that has its unit to itself runs at about 1.35 times the shared rate (measured with the 128 x 64 tile: 800 tiles
// on 512 places take 1.74 rounds, 288 tiles 0.8).
template <int KIND>
static double modelled_cost(const GemmCall& c) {
  typedef Geometry<KIND> G;
  typedef TileCost<KIND> TC;
  constexpr int kPerCu = workgroups_per_cu<KIND>();
  constexpr int kSlots = 256 * kPerCu;
  const int64_t tiles = ((c.m + G::BM - 1) / G::BM) * ((c.n + G::BN - 1) / G::BN);
  const double frac = (double)(tiles % kSlots) / kSlots;
  const double last = frac == 0.0 ? 0.0 : (kPerCu == 1 ? 1.0 : (frac <= 0.5 ? 0.75 : 0.75 + 0.5 * (frac - 0.5)));
  double best = ((double)(tiles / kSlots) + last) * (TC::t0 + TC::tk * c.k);
  int k_chunk = 0;
  double cost = 0.0;
  if (choose_splits<KIND>(c, &k_chunk, &cost) > 1) best = std::min(best, cost);
  TailPlan t;
  if (choose_tail<KIND>(c, &t, &cost)) best = std::min(best, cost);
  return best;
}

// Large or small tile?  true: the small one.  `fixed`: 1 / 2 from the environment fixes the large /
// small tile (tests, measurements).  A product that overwrites one of its operands (the triangular solves'
// B <- inv * B) relies on one large tile covering the operand's whole extent: a workgroup reads all it needs before
// it stores; memory-side accumulation is tuned for the large tile's long products.
template <int LARGE, int SMALL>
static bool small_tile_pays(const GemmCall& c, const char* variable) {
  const char* e = std::getenv(variable);  // (read per call: tests set it between calls)
  const int fixed = e ? std::atoi(e) : 0;
  if (fixed == 1) return false;
  if (fixed == 2) return true;
  const bool in_place = c.C == c.A || c.C == c.B;
  if (c.m <= 0 || c.n <= 0 || c.k <= 0 || c.atomic_c || in_place) return false;
  typedef Geometry<LARGE> GL;
  const int64_t large_tiles = ((c.m + GL::BM - 1) / GL::BM) * ((c.n + GL::BN - 1) / GL::BN);
  if (large_tiles >= 2 * 256) return false;
  return modelled_cost<SMALL>(c) < 0.95 * modelled_cost<LARGE>(c);
}

// The decisions launch_gemm_<kind> would take for this product, without launching anything:
// out[0] = 1 when the small tile is taken; out[1] = pieces the whole reduction is cut into (1: whole);
// out[2] = pieces of the reduction of the leftover tile lines of the rounds + tail form (0: that form is not used);
// out[3], out[4] = the tile's extents along m and n.
template <int KIND>
static void describe_whole(const GemmCall& c, int* out) {
  int k_chunk = 0;
  double split_cost = 0.0, tail_cost = 0.0;
  TailPlan t;
  int splits = choose_splits<KIND>(c, &k_chunk, &split_cost);
  const bool tail = choose_tail<KIND>(c, &t, &tail_cost);
  if (splits > 1 && tail && tail_cost < split_cost) splits = 1;
  out[1] = splits;
  out[2] = (splits <= 1 && tail) ? t.splits : 0;
  out[3] = Geometry<KIND>::BM;
  out[4] = Geometry<KIND>::BN;
}
template <int LARGE, int SMALL>
static void describe_plan(const GemmCall& c, const char* variable, int* out) {
  out[0] = small_tile_pays<LARGE, SMALL>(c, variable) ? 1 : 0;
  if (out[0]) describe_whole<SMALL>(c, out);
  else describe_whole<LARGE>(c, out);
}
void describe_gemm_f32(const GemmCall&, int*);
void describe_gemm_f64(const GemmCall&, int*);
void describe_gemm_c32(const GemmCall&, int*);
void describe_gemm_c64(const GemmCall&, int*);

template <int KIND, bool AK, bool BKC>
static hipError_t launch_layout(const GemmCall& c, hipStream_t stream) {
  typedef Geometry<KIND> G;
  constexpr int ES = G::ES;
  const bool aligned = ((uintptr_t)c.A % 16 == 0) && ((uintptr_t)c.B % 16 == 0) &&
                       ((c.lda * ES) % 16 == 0) && ((c.ldb * ES) % 16 == 0);
  const int64_t mi = aligned ? (c.m / G::BM) * G::BM : 0;
  const int64_t ni = aligned ? (c.n / G::BN) * G::BN : 0;
  hipError_t e;
  if (mi > 0 && ni > 0) {
    const bool bottom = c.m > mi, right = c.n > ni;
    // A few rounds of tiles only: one bounds-checked launch over everything (about 8 % slower per
    // tile than the vector-load instantiation) beats an interior launch plus edge strips whose
    // workgroups each add a whole reduction to the critical path (dgemm 4000^3: 3.0 -> 2.4 ms).
    const int64_t tiles_all = ((c.m + G::BM - 1) / G::BM) * ((c.n + G::BN - 1) / G::BN);
    if ((bottom || right) && tiles_all <= 4 * 256) return launch_whole<KIND, AK, BKC, true>(c, stream);
    if (!bottom && !right) return launch_whole<KIND, AK, BKC, false>(c, stream);
    hipStream_t sa = stream, sb = stream;
    // The strips write parts of C disjoint from the interior's and from each other's, and read
    // A and B only: the three launches may run side by side.
    const bool fork = (bottom || right) && !c.serial_edges && (!c.partials || c.partials_keep_helpers);  // (helper streams are created on first use)
    if (fork) {
      e = edge_streams_fork(stream, &sa, &sb);
      if (e != hipSuccess) return e;
    }
    e = launch_region<KIND, AK, BKC, false>(c, 0, 0, mi, ni, stream);
    if (e != hipSuccess) return e;
    e = launch_region<KIND, AK, BKC, true>(c, mi, 0, c.m - mi, c.n, sa);  // bottom strip
    if (e != hipSuccess) return e;
    e = launch_region<KIND, AK, BKC, true>(c, 0, ni, mi, c.n - ni, sb);  // right strip
    if (e != hipSuccess) return e;
    return fork ? edge_streams_join(stream, sa, sb) : hipSuccess;
  }
  return launch_whole<KIND, AK, BKC, true>(c, stream);
}

template <int KIND>
static hipError_t launch_gemm(const GemmCall& c, hipStream_t stream) {
  if (c.m <= 0 || c.n <= 0) return hipSuccess;
  if (c.k <= 0) {
    const int64_t total = c.m * c.n;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL((scale_kernel<KIND>), dim3(blocks), dim3(256), 0, stream, c.C, c.ldc,
                       (int)c.m, (int)c.n, c.beta[0], c.beta[1], c.tri);
    return hipGetLastError();
  }
  const bool ak = (c.opa == OP_T || c.opa == OP_C);    // op(A) = A^T / A^H: k runs fastest
  const bool bk = (c.opb == OP_N || c.opb == OP_CONJ);  // plain B: k runs fastest
  if (ak) return bk ? launch_layout<KIND, true, true>(c, stream) : launch_layout<KIND, true, false>(c, stream);
  return bk ? launch_layout<KIND, false, true>(c, stream) : launch_layout<KIND, false, false>(c, stream);
}

}  // namespace eigamd
