// Multi-GPU product behind the C ABI: one process drives a grid of devices.
//
// In the reference the parallel split of a product is invisible to the caller: operator*
// reaches parallelize_gemm (Eigen/src/Core/products/GeneralMatrixMatrix.h:510-511,
// Parallelizer.h:101-176), which gives every thread a column slab of C and lets all share
// one packed A panel per k-block (GeneralMatrixMatrix.h:86-158).  The multi-GPU form of
// that scheme is SUMMA on a 2-D block-cyclic distribution: C is dealt out in nb x nb blocks
// over an nprow x npcol grid, the reduction runs over k-panels of nb columns, and per panel
// the grid column that owns A's piece sends it along its grid rows, the grid row that owns
// B's piece along its grid columns, and every device multiplies its whole local C by the two
// pieces with the MFMA kernel.  K is never split, so nothing is reduced: the only exchange
// is the two panel broadcasts, issued one panel ahead of the products.
//
// Operands stay in the orientation they are stored in: for op(A) = A^T the local part of A is
// a (k-part x m-part) matrix and a panel piece is a block of its rows; the local product is
// called with the caller's op flags.  The index that op(A) shares with C (m) is always dealt
// over grid rows and the one op(B) shares with C (n) over grid columns; k is dealt over grid
// columns for A and over grid rows for B.
//
// Transports of a panel piece from its owner to the other devices of a grid row / column:
//   peer  : one copy per receiver on the receiver's copy stream (hipMemcpyPeerAsync: the DMA
//           engines move it over the xGMI link of that pair, no compute unit is involved);
//           ranks that share a device (a "virtual grid", for checking the schedule on one
//           GPU) copy device-to-device;
//   rccl  : ncclBroadcast on row / column communicators (ncclCommInitAll), all ranks of a
//           communicator in one group call.  librccl is loaded on first use.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/eigen_amd_blas.h"
#include "gemm_launch.hpp"

namespace eigamd {
int abi_fail(int code, const std::string& what);  // blas_abi.hip: sets eigamd_last_error()
int staged_copy(bool to_device, void* dev, int64_t dld, void* host, int64_t ld, int64_t rows, int64_t cols, int es, bool wait = true);
int staged_upload_wait();   // until every copy queued (wait = false) on the current device's staging handle has arrived
}
using namespace eigamd;

namespace {

constexpr int kElemBytes[4] = {4, 8, 8, 16};

#define PG_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) return abi_fail(EIGAMD_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define PG_TRY(expr)                 \
  do {                               \
    int rc_ = (expr);                \
    if (rc_ != EIGAMD_OK) return rc_; \
  } while (0)

int64_t numroc(int64_t n, int64_t nb, int p, int np) {  // entries of an n-long index, blocked by nb, that process p of np holds
  const int64_t nblocks = n / nb;
  int64_t loc = (nblocks / np) * nb;
  const int64_t extra = nblocks % np;
  if (p < extra) loc += nb;
  else if (p == extra) loc += n % nb;
  return loc;
}

// ---- RCCL, loaded on first use -------------------------------------------------------
typedef struct ncclComm* ncclComm_t;
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load(std::string* why) {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { *why = std::string("librccl not found: ") + dlerror(); return false; }
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast || !GetErrorString) {
      *why = "librccl lacks an expected symbol";
      return false;
    }
    return true;
  }
};
Rccl& rccl() {
  static Rccl r;
  return r;
}
constexpr int kNcclChar = 0;  // ncclInt8 / ncclChar

struct DeviceGuard {
  int prev = 0;
  explicit DeviceGuard(int dev) {
    (void)hipGetDevice(&prev);
    if (dev != prev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() { (void)hipSetDevice(prev); }
};

struct Local {  // one operand's local part on one rank, in stored orientation
  void* ptr = nullptr;
  int64_t rows = 0, cols = 0, ld = 1;
};

struct Rank {
  int dev = 0, pr = 0, pc = 0;
  hipStream_t compute = nullptr, comm = nullptr;
  Local a, b, c;
  void* apan[2] = {nullptr, nullptr};
  void* bpan[2] = {nullptr, nullptr};
  hipEvent_t a_ready[2] = {}, b_ready[2] = {};      // comm -> compute: slot holds its panel piece
  hipEvent_t slot_free[2] = {};                      // compute -> comm: the product that read the slot is done
  hipEvent_t a_packed[2] = {}, b_packed[2] = {};     // owner's piece is in its slot (peer transport: receivers wait)
  hipEvent_t a_fetched[2] = {}, b_fetched[2] = {};   // receiver has copied the owner's slot (owner waits before reuse)
  bool a_fetched_set[2] = {false, false}, b_fetched_set[2] = {false, false}, slot_used[2] = {false, false};
  hipEvent_t done = nullptr;
  ncclComm_t row_comm = nullptr, col_comm = nullptr;
};

}  // namespace

struct eigamd_pgemm {
  int dtype = 0, opa = 0, opb = 0, es = 8;
  int64_t m = 0, n = 0, k = 0, nb = 0;
  int nprow = 1, npcol = 1;
  bool use_rccl = false;
  bool a_rows_is_m = true, b_rows_is_k = true;
  std::vector<Rank> ranks;
  Rank& at(int pr, int pc) { return ranks[(size_t)pr * npcol + pc]; }
};

namespace {

int alloc_rank(eigamd_pgemm& g, Rank& r) {
  DeviceGuard dg(r.dev);
  PG_HIP(hipStreamCreateWithFlags(&r.compute, hipStreamNonBlocking));
  PG_HIP(hipStreamCreateWithFlags(&r.comm, hipStreamNonBlocking));
  auto ev = [&](hipEvent_t* e) { return hipEventCreateWithFlags(e, hipEventDisableTiming); };
  for (int s = 0; s < 2; ++s) {
    PG_HIP(ev(&r.a_ready[s])); PG_HIP(ev(&r.b_ready[s])); PG_HIP(ev(&r.slot_free[s]));
    PG_HIP(ev(&r.a_packed[s])); PG_HIP(ev(&r.b_packed[s]));
    PG_HIP(ev(&r.a_fetched[s])); PG_HIP(ev(&r.b_fetched[s]));
  }
  PG_HIP(ev(&r.done));
  const int64_t mloc = numroc(g.m, g.nb, r.pr, g.nprow), nloc = numroc(g.n, g.nb, r.pc, g.npcol);
  const int64_t ka = numroc(g.k, g.nb, r.pc, g.npcol), kb = numroc(g.k, g.nb, r.pr, g.nprow);
  auto set = [&](Local& l, int64_t rows, int64_t cols) -> hipError_t {
    l.rows = rows; l.cols = cols; l.ld = std::max<int64_t>((rows + 15) / 16 * 16, 16);  // 16-element leading dimensions: vector loads
    return hipMalloc(&l.ptr, (size_t)l.ld * std::max<int64_t>(cols, 1) * g.es);
  };
  PG_HIP(g.a_rows_is_m ? set(r.a, mloc, ka) : set(r.a, ka, mloc));
  PG_HIP(g.b_rows_is_k ? set(r.b, kb, nloc) : set(r.b, nloc, kb));
  PG_HIP(set(r.c, mloc, nloc));
  const int64_t ld_m = std::max<int64_t>((mloc + 15) / 16 * 16, 16), ld_n = std::max<int64_t>((nloc + 15) / 16 * 16, 16);
  const size_t abytes = (size_t)(g.a_rows_is_m ? ld_m * g.nb : g.nb * std::max<int64_t>(mloc, 1)) * g.es;
  const size_t bbytes = (size_t)(g.b_rows_is_k ? g.nb * std::max<int64_t>(nloc, 1) : ld_n * g.nb) * g.es;
  for (int s = 0; s < 2; ++s) {
    PG_HIP(hipMalloc(&r.apan[s], std::max<size_t>(abytes, 256)));
    PG_HIP(hipMalloc(&r.bpan[s], std::max<size_t>(bbytes, 256)));
  }
  return EIGAMD_OK;
}

void free_rank(Rank& r) {
  DeviceGuard dg(r.dev);
  if (r.compute) (void)hipStreamSynchronize(r.compute);
  if (r.comm) (void)hipStreamSynchronize(r.comm);
  if (r.row_comm) rccl().CommDestroy(r.row_comm);
  if (r.col_comm) rccl().CommDestroy(r.col_comm);
  for (void* p : {r.a.ptr, r.b.ptr, r.c.ptr, r.apan[0], r.apan[1], r.bpan[0], r.bpan[1]})
    if (p) (void)hipFree(p);
  for (int s = 0; s < 2; ++s)
    for (hipEvent_t e : {r.a_ready[s], r.b_ready[s], r.slot_free[s], r.a_packed[s], r.b_packed[s], r.a_fetched[s], r.b_fetched[s]})
      if (e) (void)hipEventDestroy(e);
  if (r.done) (void)hipEventDestroy(r.done);
  if (r.compute) (void)hipStreamDestroy(r.compute);
  if (r.comm) (void)hipStreamDestroy(r.comm);
}

// Geometry of a panel piece in its slot: leading dimension and bytes, for the piece of width w.
struct Piece {
  int64_t rows, cols, ld;
  size_t bytes() const { return (size_t)ld * cols; }  // elements
};
Piece a_piece(const eigamd_pgemm& g, const Rank& r, int64_t w) {
  const int64_t mloc = g.a_rows_is_m ? r.a.rows : r.a.cols;
  if (g.a_rows_is_m) return {mloc, w, std::max<int64_t>((mloc + 15) / 16 * 16, 16)};
  return {w, mloc, g.nb};
}
Piece b_piece(const eigamd_pgemm& g, const Rank& r, int64_t w) {
  const int64_t nloc = g.b_rows_is_k ? r.b.cols : r.b.rows;
  if (g.b_rows_is_k) return {w, nloc, g.nb};
  return {nloc, w, std::max<int64_t>((nloc + 15) / 16 * 16, 16)};
}

int copy_between(const Rank& dst, void* d, const Rank& src, const void* s, size_t bytes, hipStream_t q) {
  if (bytes == 0) return EIGAMD_OK;
  if (dst.dev == src.dev) PG_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, q));
  else PG_HIP(hipMemcpyPeerAsync(d, dst.dev, s, src.dev, bytes, q));
  return EIGAMD_OK;
}

// Panel p of the reduction into slot p & 1 of every rank (asynchronous).
int issue_panel(eigamd_pgemm& g, int64_t p) {
  const int s = (int)(p & 1);
  const int64_t w = std::min(g.nb, g.k - p * g.nb);
  // nobody may write a slot that a product still reads
  for (Rank& r : g.ranks) {
    DeviceGuard dg(r.dev);
    if (r.slot_used[s]) PG_HIP(hipStreamWaitEvent(r.comm, r.slot_free[s], 0));
  }
  // ---- A: the grid column p % npcol owns the piece; it travels along grid rows
  const int ca = (int)(p % g.npcol);
  const int64_t ka0 = (p / g.npcol) * g.nb;
  for (int pr = 0; pr < g.nprow; ++pr) {
    Rank& root = g.at(pr, ca);
    const Piece pc = a_piece(g, root, w);
    {
      DeviceGuard dg(root.dev);
      for (int c = 0; c < g.npcol; ++c)  // receivers of two panels ago have finished reading this slot
        if (c != ca && g.at(pr, c).a_fetched_set[s]) PG_HIP(hipStreamWaitEvent(root.comm, g.at(pr, c).a_fetched[s], 0));
      const char* src = static_cast<const char*>(root.a.ptr) + (size_t)(g.a_rows_is_m ? ka0 * root.a.ld : ka0) * g.es;
      if (pc.rows > 0 && pc.cols > 0)
        PG_HIP(hipMemcpy2DAsync(root.apan[s], (size_t)pc.ld * g.es, src, (size_t)root.a.ld * g.es, (size_t)pc.rows * g.es,
                                (size_t)pc.cols, hipMemcpyDeviceToDevice, root.comm));
      PG_HIP(hipEventRecord(root.a_packed[s], root.comm));
    }
    if ((g.npcol > 1 || g.use_rccl) && pc.bytes() > 0) {  // (RCCL also for a group of one: the same calls on every grid)
      if (g.use_rccl) {
        if (rccl().GroupStart() != 0) return abi_fail(EIGAMD_ERR_DEVICE, "ncclGroupStart failed");
        for (int c = 0; c < g.npcol; ++c) {
          Rank& r = g.at(pr, c);
          DeviceGuard dg(r.dev);
          const int rc = rccl().Broadcast(r.apan[s], r.apan[s], pc.bytes() * g.es, kNcclChar, ca, r.row_comm, r.comm);
          if (rc != 0) return abi_fail(EIGAMD_ERR_DEVICE, std::string("ncclBroadcast: ") + rccl().GetErrorString(rc));
        }
        if (rccl().GroupEnd() != 0) return abi_fail(EIGAMD_ERR_DEVICE, "ncclGroupEnd failed");
      } else {
        for (int c = 0; c < g.npcol; ++c) {
          if (c == ca) continue;
          Rank& r = g.at(pr, c);
          DeviceGuard dg(r.dev);
          PG_HIP(hipStreamWaitEvent(r.comm, root.a_packed[s], 0));
          PG_TRY(copy_between(r, r.apan[s], root, root.apan[s], pc.bytes() * g.es, r.comm));
          PG_HIP(hipEventRecord(r.a_fetched[s], r.comm));
          r.a_fetched_set[s] = true;
        }
      }
    }
  }
  // ---- B: the grid row p % nprow owns the piece; it travels along grid columns
  const int rb = (int)(p % g.nprow);
  const int64_t kb0 = (p / g.nprow) * g.nb;
  for (int c = 0; c < g.npcol; ++c) {
    Rank& root = g.at(rb, c);
    const Piece pc = b_piece(g, root, w);
    {
      DeviceGuard dg(root.dev);
      for (int pr = 0; pr < g.nprow; ++pr)
        if (pr != rb && g.at(pr, c).b_fetched_set[s]) PG_HIP(hipStreamWaitEvent(root.comm, g.at(pr, c).b_fetched[s], 0));
      const char* src = static_cast<const char*>(root.b.ptr) + (size_t)(g.b_rows_is_k ? kb0 : kb0 * root.b.ld) * g.es;
      if (pc.rows > 0 && pc.cols > 0)
        PG_HIP(hipMemcpy2DAsync(root.bpan[s], (size_t)pc.ld * g.es, src, (size_t)root.b.ld * g.es, (size_t)pc.rows * g.es,
                                (size_t)pc.cols, hipMemcpyDeviceToDevice, root.comm));
      PG_HIP(hipEventRecord(root.b_packed[s], root.comm));
    }
    if ((g.nprow > 1 || g.use_rccl) && pc.bytes() > 0) {
      if (g.use_rccl) {
        if (rccl().GroupStart() != 0) return abi_fail(EIGAMD_ERR_DEVICE, "ncclGroupStart failed");
        for (int pr = 0; pr < g.nprow; ++pr) {
          Rank& r = g.at(pr, c);
          DeviceGuard dg(r.dev);
          const int rc = rccl().Broadcast(r.bpan[s], r.bpan[s], pc.bytes() * g.es, kNcclChar, rb, r.col_comm, r.comm);
          if (rc != 0) return abi_fail(EIGAMD_ERR_DEVICE, std::string("ncclBroadcast: ") + rccl().GetErrorString(rc));
        }
        if (rccl().GroupEnd() != 0) return abi_fail(EIGAMD_ERR_DEVICE, "ncclGroupEnd failed");
      } else {
        for (int pr = 0; pr < g.nprow; ++pr) {
          if (pr == rb) continue;
          Rank& r = g.at(pr, c);
          DeviceGuard dg(r.dev);
          PG_HIP(hipStreamWaitEvent(r.comm, root.b_packed[s], 0));
          PG_TRY(copy_between(r, r.bpan[s], root, root.bpan[s], pc.bytes() * g.es, r.comm));
          PG_HIP(hipEventRecord(r.b_fetched[s], r.comm));
          r.b_fetched_set[s] = true;
        }
      }
    }
  }
  for (Rank& r : g.ranks) {
    DeviceGuard dg(r.dev);
    PG_HIP(hipEventRecord(r.a_ready[s], r.comm));  // (one stream: both pieces are in place behind this point)
    r.slot_used[s] = true;
  }
  return EIGAMD_OK;
}

hipError_t launch_kind(const GemmCall& c, hipStream_t s) {
  switch (c.kind) {
    case F32: return launch_gemm_f32(c, s);
    case F64: return launch_gemm_f64(c, s);
    case C32: return launch_gemm_c32(c, s);
    default:  return launch_gemm_c64(c, s);
  }
}

// Every rank's local product with panel p: C_loc <- alpha op(A piece) op(B piece) + beta C_loc.
int issue_products(eigamd_pgemm& g, int64_t p, const double alpha[2], const double beta[2]) {
  const int s = (int)(p & 1);
  const int64_t w = std::min(g.nb, g.k - p * g.nb);
  for (Rank& r : g.ranks) {
    DeviceGuard dg(r.dev);
    PG_HIP(hipStreamWaitEvent(r.compute, r.a_ready[s], 0));
    if (r.c.rows > 0 && r.c.cols > 0) {
      GemmCall c;
      c.kind = g.dtype; c.opa = g.opa; c.opb = g.opb;
      c.m = r.c.rows; c.n = r.c.cols; c.k = w;
      c.alpha[0] = alpha[0]; c.alpha[1] = alpha[1];
      c.beta[0] = beta[0]; c.beta[1] = beta[1];
      c.A = r.apan[s]; c.lda = a_piece(g, r, w).ld;
      c.B = r.bpan[s]; c.ldb = b_piece(g, r, w).ld;
      c.C = r.c.ptr; c.ldc = r.c.ld;
      PG_HIP(launch_kind(c, r.compute));
    }
    PG_HIP(hipEventRecord(r.slot_free[s], r.compute));
  }
  return EIGAMD_OK;
}

// Host <-> local parts, block by block through the owning device's pinned staging (block (bi, bj) of
// the stored matrix lives on the rank whose grid coordinates are the block indices modulo the grid
// extents).  `only_dev` >= 0 restricts the walk to the blocks of ranks on that device, `kpanel` >= 0 to the
// blocks of an operand (A or B) that belong to that panel of the reduction.  Downloads are synchronous (the
// local parts must be at rest); uploads return once queued unless `wait`.
int move_blocks(eigamd_pgemm& g, int which, char* host, int64_t ld, bool to_device, int only_dev = -1, int64_t kpanel = -1,
                bool wait = true) {
  // stored extents and which grid dimension deals out the stored rows / columns
  int64_t rows, cols;
  bool rows_over_grid_rows;  // stored rows dealt over grid rows (else over grid columns)
  bool k_is_rows = false;    // the reduction index runs along the stored rows (else along the stored columns)
  if (which == 0) { rows = g.a_rows_is_m ? g.m : g.k; cols = g.a_rows_is_m ? g.k : g.m; rows_over_grid_rows = g.a_rows_is_m; k_is_rows = !g.a_rows_is_m; }
  else if (which == 1) { rows = g.b_rows_is_k ? g.k : g.n; cols = g.b_rows_is_k ? g.n : g.k; rows_over_grid_rows = g.b_rows_is_k; k_is_rows = g.b_rows_is_k; }
  else { rows = g.m; cols = g.n; rows_over_grid_rows = true; }
  if (ld < std::max<int64_t>(rows, 1)) return abi_fail(EIGAMD_ERR_ARG, "leading dimension smaller than the stored rows");
  const int np_r = rows_over_grid_rows ? g.nprow : g.npcol, np_c = rows_over_grid_rows ? g.npcol : g.nprow;
  for (int64_t bj = 0; bj * g.nb < cols; ++bj) {
    if (which < 2 && kpanel >= 0 && !k_is_rows && bj != kpanel) continue;
    for (int64_t bi = 0; bi * g.nb < rows; ++bi) {
      if (which < 2 && kpanel >= 0 && k_is_rows && bi != kpanel) continue;
      const int qr = (int)(bi % np_r), qc = (int)(bj % np_c);
      Rank& r = rows_over_grid_rows ? g.at(qr, qc) : g.at(qc, qr);
      if (only_dev >= 0 && r.dev != only_dev) continue;
      Local& l = which == 0 ? r.a : (which == 1 ? r.b : r.c);
      const int64_t nr = std::min(g.nb, rows - bi * g.nb), nc = std::min(g.nb, cols - bj * g.nb);
      char* dev = static_cast<char*>(l.ptr) + (size_t)((bi / np_r) * g.nb + (bj / np_c) * g.nb * l.ld) * g.es;
      char* hst = host + (size_t)(bi * g.nb + bj * g.nb * ld) * g.es;
      DeviceGuard dg(r.dev);
      PG_TRY(staged_copy(to_device, dev, l.ld, hst, ld, nr, nc, g.es, wait));
    }
  }
  return EIGAMD_OK;
}

bool stored_rows_first(int op) { return op == OP_N || op == OP_CONJ; }

struct HostGridCache {
  std::mutex mu;
  eigamd_pgemm* g = nullptr;
  std::vector<int64_t> key;
  std::vector<int> devs;
};
HostGridCache& host_grid_cache() {
  static HostGridCache c;
  return c;
}
void release_host_grid() {
  HostGridCache& c = host_grid_cache();
  std::lock_guard<std::mutex> l(c.mu);
  if (c.g) eigamd_pgemm_destroy(c.g);
  c.g = nullptr;
}

// The copy streams start behind whatever filled the local parts, on every rank (a piece is read on its
// owner's device by every receiver).
int run_prologue(eigamd_pgemm& g) {
  for (Rank& r : g.ranks) {
    DeviceGuard dg(r.dev);
    PG_HIP(hipEventRecord(r.done, r.compute));
    PG_HIP(hipStreamWaitEvent(r.comm, r.done, 0));
  }
  if (g.ranks.size() > 1)
    for (Rank& r : g.ranks)
      for (Rank& o : g.ranks)
        if (&o != &r) {
          DeviceGuard dg(r.dev);
          PG_HIP(hipStreamWaitEvent(r.comm, o.done, 0));
        }
  return EIGAMD_OK;
}

void read_scalars(int dtype, const void* alpha, const void* beta, double al[2], double be[2]) {
  al[0] = al[1] = be[0] = be[1] = 0.0;
  switch (dtype) {
    case F32: al[0] = *static_cast<const float*>(alpha); be[0] = *static_cast<const float*>(beta); break;
    case F64: al[0] = *static_cast<const double*>(alpha); be[0] = *static_cast<const double*>(beta); break;
    case C32: al[0] = static_cast<const float*>(alpha)[0]; al[1] = static_cast<const float*>(alpha)[1];
              be[0] = static_cast<const float*>(beta)[0]; be[1] = static_cast<const float*>(beta)[1]; break;
    default:  al[0] = static_cast<const double*>(alpha)[0]; al[1] = static_cast<const double*>(alpha)[1];
              be[0] = static_cast<const double*>(beta)[0]; be[1] = static_cast<const double*>(beta)[1]; break;
  }
}

}  // namespace

extern "C" {

int eigamd_pgemm_create(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k, int nprow, int npcol, int64_t nb,
                        const int* devices, int transport, eigamd_pgemm** out) {
  if (!out) return abi_fail(EIGAMD_ERR_ARG, "null handle pointer");
  *out = nullptr;
  if (dtype < 0 || dtype > 3 || opa < 0 || opa > 3 || opb < 0 || opb > 3 || m < 0 || n < 0 || k < 0 || nprow < 1 || npcol < 1 ||
      nb < 1 || nprow * npcol > 64 || m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff || transport < 0 || transport > 1)
    return abi_fail(EIGAMD_ERR_ARG, "invalid pgemm arguments");
  if (eigamd_device_ok() != 1) return EIGAMD_ERR_NO_GPU;
  int ndev = 0;
  PG_HIP(hipGetDeviceCount(&ndev));
  eigamd_pgemm* g = new eigamd_pgemm;
  g->dtype = dtype; g->opa = opa; g->opb = opb; g->es = kElemBytes[dtype];
  g->m = m; g->n = n; g->k = k; g->nb = nb; g->nprow = nprow; g->npcol = npcol;
  g->a_rows_is_m = stored_rows_first(opa);
  g->b_rows_is_k = stored_rows_first(opb);
  g->use_rccl = transport == 1;
  g->ranks.resize((size_t)nprow * npcol);
  bool shared_device = false;
  for (int i = 0; i < nprow * npcol; ++i) {
    Rank& r = g->ranks[i];
    r.pr = i / npcol; r.pc = i % npcol;
    r.dev = devices ? devices[i] : i % ndev;
    if (r.dev < 0 || r.dev >= ndev) { delete g; return abi_fail(EIGAMD_ERR_ARG, "device ordinal out of range"); }
    for (int j = 0; j < i; ++j) shared_device = shared_device || g->ranks[j].dev == r.dev;
  }
  if (g->use_rccl && shared_device) { delete g; return abi_fail(EIGAMD_ERR_ARG, "the RCCL transport needs one device per rank"); }
  int rc = EIGAMD_OK;
  for (Rank& r : g->ranks)
    if ((rc = alloc_rank(*g, r)) != EIGAMD_OK) break;
  if (rc == EIGAMD_OK && !shared_device)
    for (Rank& r : g->ranks) {  // direct copies between the devices of the grid
      DeviceGuard dg(r.dev);
      for (Rank& o : g->ranks)
        if (o.dev != r.dev) {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, r.dev, o.dev) == hipSuccess && can) {
            (void)hipDeviceEnablePeerAccess(o.dev, 0);  // (already enabled, or not possible: the copies then go through the host)
            (void)hipGetLastError();
          }
        }
    }
  if (rc == EIGAMD_OK && g->use_rccl) {
    std::string why;
    if (!rccl().load(&why)) rc = abi_fail(EIGAMD_ERR_DEVICE, why);
    for (int pr = 0; rc == EIGAMD_OK && pr < nprow; ++pr) {
      std::vector<int> devs(npcol);
      std::vector<ncclComm_t> comms(npcol);
      for (int c = 0; c < npcol; ++c) devs[c] = g->at(pr, c).dev;
      const int e = rccl().CommInitAll(comms.data(), npcol, devs.data());
      if (e != 0) rc = abi_fail(EIGAMD_ERR_DEVICE, std::string("ncclCommInitAll: ") + rccl().GetErrorString(e));
      else for (int c = 0; c < npcol; ++c) g->at(pr, c).row_comm = comms[c];
    }
    for (int c = 0; rc == EIGAMD_OK && c < npcol; ++c) {
      std::vector<int> devs(nprow);
      std::vector<ncclComm_t> comms(nprow);
      for (int pr = 0; pr < nprow; ++pr) devs[pr] = g->at(pr, c).dev;
      const int e = rccl().CommInitAll(comms.data(), nprow, devs.data());
      if (e != 0) rc = abi_fail(EIGAMD_ERR_DEVICE, std::string("ncclCommInitAll: ") + rccl().GetErrorString(e));
      else for (int pr = 0; pr < nprow; ++pr) g->at(pr, c).col_comm = comms[pr];
    }
  }
  if (rc != EIGAMD_OK) {
    for (Rank& r : g->ranks) free_rank(r);
    delete g;
    return rc;
  }
  *out = g;
  return EIGAMD_OK;
}

int eigamd_pgemm_destroy(eigamd_pgemm* g) {
  if (!g) return EIGAMD_OK;
  for (Rank& r : g->ranks) free_rank(r);
  delete g;
  return EIGAMD_OK;
}

int eigamd_pgemm_fill(eigamd_pgemm* g, int which, uint64_t seed) {
  if (!g || which < 0 || which > 2) return abi_fail(EIGAMD_ERR_ARG, "invalid pgemm_fill arguments");
  for (Rank& r : g->ranks) {
    DeviceGuard dg(r.dev);
    Local& l = which == 0 ? r.a : (which == 1 ? r.b : r.c);
    if (l.rows == 0 || l.cols == 0) continue;
    // which grid coordinate deals out the stored rows / columns of this operand
    const bool rows_over_grid_rows = which == 0 ? g->a_rows_is_m : (which == 1 ? g->b_rows_is_k : true);
    const int np_r = rows_over_grid_rows ? g->nprow : g->npcol, p_r = rows_over_grid_rows ? r.pr : r.pc;
    const int np_c = rows_over_grid_rows ? g->npcol : g->nprow, p_c = rows_over_grid_rows ? r.pc : r.pr;
    PG_TRY(eigamd_fill_uniform_cyclic_dev(g->dtype, l.ptr, l.rows, l.cols, l.ld, seed, g->nb, np_r, p_r, g->nb, np_c, p_c, r.compute));
  }
  return EIGAMD_OK;
}

int eigamd_pgemm_scatter(eigamd_pgemm* g, int which, const void* host, int64_t ld) {
  if (!g || which < 0 || which > 2 || !host) return abi_fail(EIGAMD_ERR_ARG, "invalid pgemm_scatter arguments");
  PG_TRY(eigamd_pgemm_sync(g));
  return move_blocks(*g, which, const_cast<char*>(static_cast<const char*>(host)), ld, true);
}

int eigamd_pgemm_gather(eigamd_pgemm* g, int which, void* host, int64_t ld) {
  if (!g || which < 0 || which > 2 || !host) return abi_fail(EIGAMD_ERR_ARG, "invalid pgemm_gather arguments");
  PG_TRY(eigamd_pgemm_sync(g));
  return move_blocks(*g, which, static_cast<char*>(host), ld, false);
}

int eigamd_pgemm_local(eigamd_pgemm* g, int which, int rank, void** ptr, int64_t* rows, int64_t* cols, int64_t* ld, int* device) {
  if (!g || which < 0 || which > 2 || rank < 0 || rank >= (int)g->ranks.size()) return abi_fail(EIGAMD_ERR_ARG, "invalid pgemm_local arguments");
  Rank& r = g->ranks[rank];
  Local& l = which == 0 ? r.a : (which == 1 ? r.b : r.c);
  if (ptr) *ptr = l.ptr;
  if (rows) *rows = l.rows;
  if (cols) *cols = l.cols;
  if (ld) *ld = l.ld;
  if (device) *device = r.dev;
  return EIGAMD_OK;
}

int eigamd_pgemm_run(eigamd_pgemm* g, const void* alpha, const void* beta) {
  if (!g || !alpha || !beta) return abi_fail(EIGAMD_ERR_ARG, "invalid pgemm_run arguments");
  double al[2], be[2];
  read_scalars(g->dtype, alpha, beta, al, be);
  const int64_t npanels = (g->k + g->nb - 1) / g->nb;
  const double one[2] = {1.0, 0.0};
  if (npanels == 0) {  // k == 0: C <- beta C (blas/level3_impl.h:62-69)
    if (be[0] == 1.0 && be[1] == 0.0) return EIGAMD_OK;
    for (Rank& r : g->ranks) {
      DeviceGuard dg(r.dev);
      if (r.c.rows == 0 || r.c.cols == 0) continue;
      GemmCall c;
      c.kind = g->dtype; c.opa = OP_N; c.opb = OP_N; c.m = r.c.rows; c.n = r.c.cols; c.k = 0;
      c.alpha[0] = c.alpha[1] = 0.0; c.beta[0] = be[0]; c.beta[1] = be[1];
      c.A = nullptr; c.lda = 1; c.B = nullptr; c.ldb = 1; c.C = r.c.ptr; c.ldc = r.c.ld;
      PG_HIP(launch_kind(c, r.compute));
    }
    return EIGAMD_OK;
  }
  PG_TRY(run_prologue(*g));
  PG_TRY(issue_panel(*g, 0));
  for (int64_t p = 0; p < npanels; ++p) {
    if (p + 1 < npanels) PG_TRY(issue_panel(*g, p + 1));  // one panel ahead of the products
    PG_TRY(issue_products(*g, p, al, p == 0 ? be : one));
  }
  return EIGAMD_OK;
}

int eigamd_pgemm_sync(eigamd_pgemm* g) {
  if (!g) return EIGAMD_OK;
  for (Rank& r : g->ranks) {
    DeviceGuard dg(r.dev);
    PG_HIP(hipStreamSynchronize(r.compute));
    PG_HIP(hipStreamSynchronize(r.comm));
  }
  return EIGAMD_OK;
}

// One call from host memory to host memory.  The route is bound by the host links, so everything that can
// overlap does: every device of the grid has a host thread of its own that moves that device's blocks through
// that device's pinned staging (one PCIe link each); the operands go up in the order the reduction consumes
// them -- panel 0 of A and B, then C (not at all when beta == 0: the first product overwrites), then the other
// panels -- and the calling thread issues the broadcast and the products of panel p as soon as every device has
// reported panel p landed, so the products run underneath the upload of the later panels; when its last product
// has been queued every device thread waits for it and brings its own blocks of C back.
int eigamd_pgemm_host(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k, const void* alpha, const void* a,
                      int64_t lda, const void* b, int64_t ldb, const void* beta, void* c, int64_t ldc, int nprow, int npcol,
                      int64_t nb, const int* devices, int transport) {
  if (m == 0 || n == 0) return EIGAMD_OK;
  if (!alpha || !beta || !c || (k > 0 && (!a || !b))) return abi_fail(EIGAMD_ERR_ARG, "invalid pgemm_host arguments");
  const bool trace = getenv("EIGAMD_PGEMM_TRACE") != nullptr;   // where the wall time of the call goes, on stderr
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
  // One call at a time; the grid of the last call is kept (its streams, events and device memory cost about
  // 12 ms to set up and tear down at 8192^3, a sixth of the call) and serves the next call of the same shape.
  HostGridCache& cache = host_grid_cache();
  std::lock_guard<std::mutex> cache_lock(cache.mu);
  std::vector<int> key_devs;
  for (int i = 0; i < nprow * npcol && devices; ++i) key_devs.push_back(devices[i]);
  const std::vector<int64_t> key = {dtype, opa, opb, m, n, k, nprow, npcol, nb, transport};
  if (cache.g && (cache.key != key || cache.devs != key_devs)) {
    eigamd_pgemm_destroy(cache.g);
    cache.g = nullptr;
  }
  if (!cache.g) {
    PG_TRY(eigamd_pgemm_create(dtype, opa, opb, m, n, k, nprow, npcol, nb, devices, transport, &cache.g));
    cache.key = key;
    cache.devs = key_devs;
    static const int registered = std::atexit(release_host_grid);
    (void)registered;
  }
  eigamd_pgemm* g = cache.g;
  for (Rank& r : g->ranks)   // a fresh run: no slot holds a piece, nobody is reading one
    for (int s2 = 0; s2 < 2; ++s2) r.slot_used[s2] = r.a_fetched_set[s2] = r.b_fetched_set[s2] = false;
  auto drop_grid = [&]() {   // after a failure: the next call starts from a new grid
    (void)eigamd_pgemm_sync(g);
    eigamd_pgemm_destroy(g);
    cache.g = nullptr;
  };
  double al[2], be[2];
  read_scalars(dtype, alpha, beta, al, be);
  const bool beta_is_zero = be[0] == 0.0 && be[1] == 0.0;
  const int64_t npanels = (k + nb - 1) / nb;
  int rc = EIGAMD_OK;
  if (npanels == 0) {  // nothing to multiply: C <- beta C
    if (!beta_is_zero) rc = eigamd_pgemm_scatter(g, 2, c, ldc);
    if (rc == EIGAMD_OK) rc = eigamd_pgemm_run(g, alpha, beta);
    if (rc == EIGAMD_OK) rc = eigamd_pgemm_sync(g);
    if (rc == EIGAMD_OK) rc = eigamd_pgemm_gather(g, 2, c, ldc);
    if (rc != EIGAMD_OK) drop_grid();
    return rc;
  }
  std::vector<int> devs;
  for (const Rank& r : g->ranks)
    if (std::find(devs.begin(), devs.end(), r.dev) == devs.end()) devs.push_back(r.dev);
  struct Shared {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> landed;   // landed[p]: devices whose blocks of panel p (and, with p == 0, of C) are in device memory
    bool issued = false;       // every product has been queued and Rank::done recorded behind the last one
    int rc = EIGAMD_OK;        // first failure of any thread
    std::string what;          // ... and its text (eigamd_last_error() is per thread: the caller's thread repeats it)
  } sh;
  sh.landed.assign((size_t)npanels, 0);
  char* ah = const_cast<char*>(static_cast<const char*>(a));
  char* bh = const_cast<char*>(static_cast<const char*>(b));
  char* ch = static_cast<char*>(c);
  auto fail_with = [&](int code) {
    std::lock_guard<std::mutex> l(sh.mu);
    if (sh.rc == EIGAMD_OK) {
      sh.rc = code;
      sh.what = eigamd_last_error();
    }
    sh.cv.notify_all();
  };
  auto device_thread = [&](int dev) {
    if (hipSetDevice(dev) != hipSuccess) { fail_with(abi_fail(EIGAMD_ERR_DEVICE, "hipSetDevice failed in a copy thread")); return; }
    for (int64_t p = 0; p < npanels; ++p) {
      int e = move_blocks(*g, 0, ah, lda, true, dev, p, false);
      if (e == EIGAMD_OK) e = move_blocks(*g, 1, bh, ldb, true, dev, p, false);
      if (e == EIGAMD_OK && p == 0 && !beta_is_zero) e = move_blocks(*g, 2, ch, ldc, true, dev, -1, false);
      if (e == EIGAMD_OK) e = staged_upload_wait();
      if (e != EIGAMD_OK) { fail_with(e); return; }
      {
        std::lock_guard<std::mutex> l(sh.mu);
        if (sh.rc != EIGAMD_OK) return;
        ++sh.landed[(size_t)p];
      }
      sh.cv.notify_all();
      if (trace) fprintf(stderr, "pgemm_host: device %d panel %lld landed at %.2f ms\n", dev, (long long)p, since());
    }
    {
      std::unique_lock<std::mutex> l(sh.mu);
      sh.cv.wait(l, [&] { return sh.issued || sh.rc != EIGAMD_OK; });
      if (sh.rc != EIGAMD_OK) return;
    }
    for (Rank& r : g->ranks)
      if (r.dev == dev && hipEventSynchronize(r.done) != hipSuccess) {
        fail_with(abi_fail(EIGAMD_ERR_DEVICE, "waiting for a device's last product failed"));
        return;
      }
    if (trace) fprintf(stderr, "pgemm_host: device %d products done at %.2f ms\n", dev, since());
    int e = move_blocks(*g, 2, ch, ldc, false, dev, -1, false);
    if (e == EIGAMD_OK) e = staged_upload_wait();
    if (e != EIGAMD_OK) fail_with(e);
    if (trace) fprintf(stderr, "pgemm_host: device %d C back at %.2f ms\n", dev, since());
  };
  if (trace) fprintf(stderr, "pgemm_host: grid %dx%d created at %.2f ms\n", nprow, npcol, since());
  std::vector<std::thread> threads;
  const double one[2] = {1.0, 0.0};
  try {
    for (int d : devs) threads.emplace_back(device_thread, d);
  } catch (const std::system_error& e) {   // (no thread to be had: the ones that did start are told to stop)
    rc = abi_fail(EIGAMD_ERR_DEVICE, std::string("cannot start a copy thread: ") + e.what());
    fail_with(rc);
  }
  if (rc == EIGAMD_OK) rc = run_prologue(*g);
  for (int64_t p = 0; rc == EIGAMD_OK && p < npanels; ++p) {
    {
      std::unique_lock<std::mutex> l(sh.mu);
      sh.cv.wait(l, [&] { return sh.landed[(size_t)p] == (int)devs.size() || sh.rc != EIGAMD_OK; });
      if (sh.rc != EIGAMD_OK) break;
    }
    rc = issue_panel(*g, p);
    if (rc == EIGAMD_OK) rc = issue_products(*g, p, al, p == 0 ? be : one);
  }
  if (rc == EIGAMD_OK)
    for (Rank& r : g->ranks) {
      DeviceGuard dg(r.dev);
      if (hipEventRecord(r.done, r.compute) != hipSuccess) { rc = abi_fail(EIGAMD_ERR_DEVICE, "hipEventRecord failed"); break; }
    }
  if (rc != EIGAMD_OK) fail_with(rc);
  {
    std::lock_guard<std::mutex> l(sh.mu);
    sh.issued = true;
  }
  sh.cv.notify_all();
  for (std::thread& t : threads) t.join();
  if (rc == EIGAMD_OK && sh.rc != EIGAMD_OK) rc = abi_fail(sh.rc, sh.what);
  if (rc == EIGAMD_OK) rc = eigamd_pgemm_sync(g);
  if (rc != EIGAMD_OK) drop_grid();
  if (trace) fprintf(stderr, "pgemm_host: done at %.2f ms\n", since());
  return rc;
}

}  // extern "C"
