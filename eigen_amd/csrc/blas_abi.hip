// C ABI of libeigen_amd_blas.so (declarations: include/eigen_amd_blas.h).
//
// The F77 entry points restate the argument handling of EIGEN_BLAS_FUNC(gemm)
// (blas/level3_impl.h:47-69 of the Eigen tree) and hand the product itself to
// the MFMA kernels; there is no CPU compute path in this library.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/eigen_amd_blas.h"
#include "gemm_launch.hpp"
#include "lu.hpp"
#include "trsm.hpp"
#include "syrk.hpp"
#include "level3x.hpp"
#include "rng.hpp"

using namespace eigamd;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& what) {
  g_err = what;
  return code;
}
int hip_fail(hipError_t e, const char* where) {
  return fail(EIGAMD_ERR_DEVICE, std::string(where) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);   \
  } while (0)

constexpr int kElemSize[4] = {4, 8, 8, 16};

hipError_t launch(const GemmCall& c, hipStream_t s) {
  switch (c.kind) {
    case F32: return launch_gemm_f32(c, s);
    case F64: return launch_gemm_f64(c, s);
    case C32: return launch_gemm_c32(c, s);
    case C64: return launch_gemm_c64(c, s);
  }
  return hipErrorInvalidValue;
}

void read_scalar(int dtype, const void* p, double out[2]) {
  switch (dtype) {
    case F32: out[0] = *static_cast<const float*>(p); out[1] = 0; break;
    case F64: out[0] = *static_cast<const double*>(p); out[1] = 0; break;
    case C32: out[0] = static_cast<const float*>(p)[0]; out[1] = static_cast<const float*>(p)[1]; break;
    default:  out[0] = static_cast<const double*>(p)[0]; out[1] = static_cast<const double*>(p)[1]; break;
  }
}

bool stored_rows_is_m(int op) { return op == OP_N || op == OP_CONJ; }

// Argument checks shared by the 64-bit entry points; the order and the codes
// are those of blas/level3_impl.h:47-55.
int check_args(int opa, int opb, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb,
               int64_t ldc) {
  if (opa < 0 || opa > 3) return 1;
  if (opb < 0 || opb > 3) return 2;
  if (m < 0) return 3;
  if (n < 0) return 4;
  if (k < 0) return 5;
  if (lda < std::max<int64_t>(1, stored_rows_is_m(opa) ? m : k)) return 8;
  if (ldb < std::max<int64_t>(1, stored_rows_is_m(opb) ? k : n)) return 10;
  if (ldc < std::max<int64_t>(1, m)) return 13;
  return 0;
}

int ensure_device() {
  static std::atomic<int> state_cache{0};  // 0 unknown, 1 ok, -1 none
  int state = state_cache.load();
  if (state == 0) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
      state = -1;
    } else {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
          std::strncmp(prop.gcnArchName, "gfx950", 6) == 0)
        state = 1;
      else
        state = -1;
    }
    state_cache.store(state);
  }
  if (state < 0) return fail(EIGAMD_ERR_NO_GPU, "no gfx950 (MI355X) device is visible; this library has no CPU path");
  return EIGAMD_OK;
}

// A few host threads that copy between the caller's (pageable) matrices and pinned
// staging memory: one thread moves ~10 GB/s, PCIe moves five times that.
class CopyPool {
 public:
  explicit CopyPool(int nthreads) {
    for (int t = 0; t < nthreads; ++t) workers_.emplace_back([this] { loop(); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  // fn(i) for i in [0, n), spread over the workers and the calling thread; returns when all ran.
  void parallel_for(int64_t n, const std::function<void(int64_t)>& fn) {
    if (n <= 0) return;
    if (workers_.empty() || n == 1) {
      for (int64_t i = 0; i < n; ++i) fn(i);
      return;
    }
    {
      std::lock_guard<std::mutex> l(mu_);
      fn_ = &fn;
      n_ = n;
      next_ = 0;
      active_ = (int)workers_.size();
      ++generation_;
    }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> l(mu_);
    done_cv_.wait(l, [this] { return active_ == 0; });
    fn_ = nullptr;
  }

 private:
  void drain() {
    for (;;) {
      const int64_t i = next_.fetch_add(1);
      if (i >= n_) break;
      (*fn_)(i);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      drain();
      {
        std::lock_guard<std::mutex> l(mu_);
        --active_;
      }
      done_cv_.notify_all();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int64_t)>* fn_ = nullptr;
  int64_t n_ = 0;
  std::atomic<int64_t> next_{0};
  int active_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

// Per-process handle of the host-pointer entry points: grow-only device buffers, pinned
// staging chunks (two per direction), three streams, the copy threads.  Guarded by one
// mutex (callers may invoke ?gemm_ from several host threads; the reference entry points
// are re-entrant).
struct Staging {
  static constexpr size_t kChunk = 64u << 20;
  std::mutex mu;
  void* buf[3] = {nullptr, nullptr, nullptr};
  size_t cap[3] = {0, 0, 0};
  hipStream_t stream = nullptr;   // compute
  hipStream_t h2d = nullptr, d2h = nullptr;
  void* up[2] = {nullptr, nullptr};
  void* down[2] = {nullptr, nullptr};
  hipEvent_t up_free[2] = {nullptr, nullptr};     // the DMA out of up[i] has finished
  hipEvent_t down_ready[2] = {nullptr, nullptr};  // the DMA into down[i] has finished
  hipEvent_t uploaded = nullptr, computed = nullptr, computed2 = nullptr;  // computed / computed2: the two column panels in flight
  int up_next = 0, down_next = 0;
  CopyPool* pool = nullptr;
  // Blocks on their way back that download_queue has started and not yet copied out of their pinned chunk.
  struct PendingDown {
    bool active = false;
    char* host = nullptr;
    int64_t ld = 0, rows = 0, cols = 0;
    int es = 0;
  } pending[2];
  int reserve(int i, size_t bytes) {
    if (bytes <= cap[i]) return EIGAMD_OK;
    if (buf[i]) { HIP_TRY(hipFree(buf[i])); buf[i] = nullptr; cap[i] = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipMalloc(&buf[i], want));
    cap[i] = want;
    return EIGAMD_OK;
  }
  int init() {
    if (stream) return EIGAMD_OK;
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&h2d, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&d2h, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIP_TRY(hipHostMalloc(&up[i], kChunk, hipHostMallocDefault));
      HIP_TRY(hipHostMalloc(&down[i], kChunk, hipHostMallocDefault));
      HIP_TRY(hipEventCreateWithFlags(&up_free[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&down_ready[i], hipEventDisableTiming));
      HIP_TRY(hipEventRecord(up_free[i], h2d));
    }
    HIP_TRY(hipEventCreateWithFlags(&uploaded, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&computed, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&computed2, hipEventDisableTiming));
    int nthreads = 8;
    if (const char* e = getenv("EIGAMD_HOST_THREADS")) nthreads = std::max(1, atoi(e));
    pool = new CopyPool(nthreads - 1);
    return EIGAMD_OK;
  }

  // rows x cols elements of `es` bytes from host (leading dimension ld) to device (dld),
  // through the pinned chunks, asynchronously on the h2d stream.
  // `inc` is the distance, in elements, between consecutive rows of a host column (1 for an
  // ordinary column-major matrix); the device image is always dense along a column.
  int upload(void* dev, int64_t dld, const void* host, int64_t ld, int64_t rows, int64_t cols, int es, int64_t inc = 1) {
    if (rows <= 0 || cols <= 0) return EIGAMD_OK;
    const size_t colbytes = (size_t)rows * es;
    if (colbytes > kChunk) {  // a single column does not fit a chunk: let the runtime stage it
      if (inc != 1) return fail(EIGAMD_ERR_ARG, "a strided column above 64 MiB is not supported");
      HIP_TRY(hipMemcpy2DAsync(dev, dld * es, host, ld * es, colbytes, cols, hipMemcpyHostToDevice, h2d));
      return EIGAMD_OK;
    }
    const int64_t per = std::max<int64_t>(1, (int64_t)(kChunk / colbytes));
    for (int64_t c0 = 0; c0 < cols; c0 += per) {
      const int64_t nc = std::min(per, cols - c0);
      const int i = up_next;
      up_next ^= 1;
      HIP_TRY(hipEventSynchronize(up_free[i]));
      char* pin = static_cast<char*>(up[i]);
      const char* src = static_cast<const char*>(host) + (size_t)c0 * ld * es;
      if (inc == 1) copy_columns(pin, colbytes, src, (size_t)ld * es, colbytes, nc);
      else          copy_strided(pin, colbytes, 1, src, (size_t)ld * es, inc, rows, nc, es);
      HIP_TRY(hipMemcpy2DAsync(static_cast<char*>(dev) + (size_t)c0 * dld * es, dld * es, pin, colbytes, colbytes, nc,
                               hipMemcpyHostToDevice, h2d));
      HIP_TRY(hipEventRecord(up_free[i], h2d));
    }
    return EIGAMD_OK;
  }

  // A block that fits one pinned chunk starts its way back and the call returns: the DMA of this block runs while
  // the threads copy the block queued before it out of the other chunk (a walk over many blocks that waited for
  // each one paid a DMA and a copy per block, one after the other).  download_flush() brings the last ones home.
  int download_finish(int i) {
    PendingDown& p = pending[i];
    if (!p.active) return EIGAMD_OK;
    HIP_TRY(hipEventSynchronize(down_ready[i]));
    const size_t colbytes = (size_t)p.rows * p.es;
    copy_columns(p.host, (size_t)p.ld * p.es, static_cast<const char*>(down[i]), colbytes, colbytes, p.cols);
    p.active = false;
    return EIGAMD_OK;
  }
  int download_flush() {
    int rc = download_finish(down_next);       // the older of the two
    if (rc == EIGAMD_OK) rc = download_finish(down_next ^ 1);
    return rc;
  }
  int download_queue(void* host, int64_t ld, const void* dev, int64_t dld, int64_t rows, int64_t cols, int es) {
    if (rows <= 0 || cols <= 0) return EIGAMD_OK;
    const size_t colbytes = (size_t)rows * es;
    int rc;
    if (colbytes * (size_t)cols > kChunk) {
      if ((rc = download_flush()) != EIGAMD_OK) return rc;
      return download(host, ld, dev, dld, rows, cols, es);
    }
    const int i = down_next;
    down_next ^= 1;
    if ((rc = download_finish(i)) != EIGAMD_OK) return rc;
    HIP_TRY(hipMemcpy2DAsync(down[i], colbytes, dev, dld * es, colbytes, cols, hipMemcpyDeviceToHost, d2h));
    HIP_TRY(hipEventRecord(down_ready[i], d2h));
    pending[i].active = true;
    pending[i].host = static_cast<char*>(host);
    pending[i].ld = ld; pending[i].rows = rows; pending[i].cols = cols; pending[i].es = es;
    return download_finish(i ^ 1);
  }

  // The reverse, on the d2h stream; returns when the data is in the caller's memory.  The DMA
  // of chunk c+1 runs while the threads copy chunk c out of its pinned buffer.
  int download(void* host, int64_t ld, const void* dev, int64_t dld, int64_t rows, int64_t cols, int es, int64_t inc = 1) {
    if (rows <= 0 || cols <= 0) return EIGAMD_OK;
    {
      const int rc0 = download_flush();   // (the chunks are about to be reused)
      if (rc0 != EIGAMD_OK) return rc0;
    }
    const size_t colbytes = (size_t)rows * es;
    if (colbytes > kChunk) {
      if (inc != 1) return fail(EIGAMD_ERR_ARG, "a strided column above 64 MiB is not supported");
      HIP_TRY(hipMemcpy2DAsync(host, ld * es, dev, dld * es, colbytes, cols, hipMemcpyDeviceToHost, d2h));
      HIP_TRY(hipStreamSynchronize(d2h));
      return EIGAMD_OK;
    }
    const int64_t per = std::max<int64_t>(1, (int64_t)(kChunk / colbytes));
    auto issue = [&](int64_t c0, int i) -> int {
      const int64_t nc = std::min(per, cols - c0);
      HIP_TRY(hipMemcpy2DAsync(down[i], colbytes, static_cast<const char*>(dev) + (size_t)c0 * dld * es, dld * es, colbytes,
                               nc, hipMemcpyDeviceToHost, d2h));
      HIP_TRY(hipEventRecord(down_ready[i], d2h));
      return EIGAMD_OK;
    };
    int rc = issue(0, 0);
    if (rc != EIGAMD_OK) return rc;
    int cur = 0;
    for (int64_t c0 = 0; c0 < cols; c0 += per) {
      const int64_t nc = std::min(per, cols - c0);
      if (c0 + per < cols && (rc = issue(c0 + per, cur ^ 1)) != EIGAMD_OK) return rc;
      HIP_TRY(hipEventSynchronize(down_ready[cur]));
      if (inc == 1)
        copy_columns(static_cast<char*>(host) + (size_t)c0 * ld * es, (size_t)ld * es, static_cast<const char*>(down[cur]),
                     colbytes, colbytes, nc);
      else
        copy_strided(static_cast<char*>(host) + (size_t)c0 * ld * es, (size_t)ld * es, inc,
                     static_cast<const char*>(down[cur]), colbytes, 1, rows, nc, es);
      cur ^= 1;
    }
    return EIGAMD_OK;
  }

 private:
  // ncols columns of `bytes` bytes each; split into pieces so that all threads have work.
  void copy_columns(char* dst, size_t dst_pitch, const char* src, size_t src_pitch, size_t bytes, int64_t ncols) {
    const size_t piece = 1u << 20;
    const int64_t per_col = (int64_t)((bytes + piece - 1) / piece);
    pool->parallel_for(ncols * per_col, [&](int64_t t) {
      const int64_t c = t / per_col, q = t % per_col;
      const size_t off = (size_t)q * piece, len = std::min(piece, bytes - off);
      std::memcpy(dst + (size_t)c * dst_pitch + off, src + (size_t)c * src_pitch + off, len);
    });
  }
  // The same for columns whose rows lie `dst_inc` / `src_inc` elements apart (a destination
  // with a non-unit inner stride is gathered into, and scattered out of, the dense staging image).
  template <typename E>
  static void strided_piece(char* dst, int64_t dst_inc, const char* src, int64_t src_inc, int64_t count) {
    E* d = reinterpret_cast<E*>(dst);
    const E* s = reinterpret_cast<const E*>(src);
    for (int64_t i = 0; i < count; ++i) d[i * dst_inc] = s[i * src_inc];
  }
  void copy_strided(char* dst, size_t dst_pitch, int64_t dst_inc, const char* src, size_t src_pitch, int64_t src_inc,
                    int64_t rows, int64_t ncols, int es) {
    const int64_t piece = 1 << 16;  // elements
    const int64_t per_col = (rows + piece - 1) / piece;
    struct E16 { uint64_t a, b; };
    pool->parallel_for(ncols * per_col, [&](int64_t t) {
      const int64_t c = t / per_col, r0 = (t % per_col) * piece, cnt = std::min(piece, rows - r0);
      char* d = dst + (size_t)c * dst_pitch + (size_t)r0 * dst_inc * es;
      const char* sp = src + (size_t)c * src_pitch + (size_t)r0 * src_inc * es;
      if (es == 4) strided_piece<uint32_t>(d, dst_inc, sp, src_inc, cnt);
      else if (es == 8) strided_piece<uint64_t>(d, dst_inc, sp, src_inc, cnt);
      else strided_piece<E16>(d, dst_inc, sp, src_inc, cnt);
    });
  }
};
Staging& staging() {
  static Staging s[kMaxDevices];
  return s[current_device_slot()];
}

int64_t round_up(int64_t v, int64_t q) { return (v + q - 1) / q * q; }

// When a host-pointer product is dealt out over several devices (read at every call: cheap, and
// tests change the variables between calls).  The route is taken on request only -- EIGAMD_MULTI_GPU=1:
// every visible device, or EIGAMD_MULTI_GPU_VIRTUAL=P (tests): P ranks on the current device -- because a
// grid allocates memory on every device it spans, and a process that is one of several sharing the node
// must not spread over its neighbours' devices: a launcher that gives every device its own process is
// recognised by its environment (WORLD_SIZE, LOCAL_WORLD_SIZE, OMPI_COMM_WORLD_SIZE, PMI_SIZE, SLURM_NTASKS
// above 1) and wins over the request.
struct MultiGpuRule {
  int ranks = 1;
  bool virtual_grid = false;
  int64_t nb = 2048;
  double min_work = 8192.0 * 8192.0 * 8192.0;
};
MultiGpuRule multi_gpu_rule() {
  MultiGpuRule r;
  if (const char* e = getenv("EIGAMD_MULTI_GPU_VIRTUAL")) {
    r.ranks = std::max(1, atoi(e));
    r.virtual_grid = true;
  } else {
    const char* on = getenv("EIGAMD_MULTI_GPU");
    if (!on || atoi(on) == 0) return r;
    for (const char* name : {"WORLD_SIZE", "LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS"})
      if (const char* v = getenv(name))
        if (atoi(v) > 1) return r;
    int count = 1;
    if (hipGetDeviceCount(&count) != hipSuccess) count = 1;
    r.ranks = std::min(count, 8);
  }
  if (const char* e = getenv("EIGAMD_MULTI_GPU_NB")) r.nb = std::max<int64_t>(16, atoll(e));
  if (const char* e = getenv("EIGAMD_MULTI_GPU_MIN_WORK")) r.min_work = atof(e);
  return r;
}

// Small products (every operand fits a fraction of one pinned chunk): the general path below wakes the
// copy threads, issues a 2-D copy per operand on three streams and waits on events -- 130 us before any
// work is done.  Here the calling thread packs A, B and (unless beta == 0) C into one pinned chunk, ONE
// copy takes them up, the kernel runs, ONE copy brings C back, one wait.  There is still no CPU product:
// a 32^3 dgemm_ costs about 50 us where Eigen's own kernel needs 1 us (no fallback, by design).
// Every call borrows a context of its own (a stream, two pinned chunks, a device buffer) from a per-device
// pool that grows to the number of calls ever in flight together: host threads that multiply at the same time
// (Eigen products inside the caller's own parallel region) do not wait for each other, as the reference's
// ?gemm_ is re-entrant (blas/level3_impl.h:12-76 keeps no state between calls).
constexpr size_t kSmallProductBytes = (size_t)3 << 20;
struct SmallContext {
  hipStream_t stream = nullptr;
  char* up = nullptr;    // pinned: A, B, C on their way to the device
  char* down = nullptr;  // pinned: C on its way back
  char* dev = nullptr;
};
constexpr size_t kSmallContextBytes = kSmallProductBytes + 4096;  // (+ the 256-byte alignment of the three images)
struct SmallPool {
  std::mutex mu;
  std::vector<SmallContext*> idle;
};
SmallPool& small_pool() {
  static SmallPool p[kMaxDevices];
  return p[current_device_slot()];
}
int small_context_acquire(SmallContext** out) {
  SmallPool& pool = small_pool();
  {
    std::lock_guard<std::mutex> lock(pool.mu);
    if (!pool.idle.empty()) {
      *out = pool.idle.back();
      pool.idle.pop_back();
      return EIGAMD_OK;
    }
  }
  SmallContext* c = new SmallContext;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->up), kSmallContextBytes, hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->down), kSmallContextBytes, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->dev), kSmallContextBytes);
  if (e != hipSuccess) {
    if (c->dev) (void)hipFree(c->dev);
    if (c->down) (void)hipHostFree(c->down);
    if (c->up) (void)hipHostFree(c->up);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return hip_fail(e, "context of a small host-pointer product");
  }
  *out = c;
  return EIGAMD_OK;
}
struct SmallContextLease {  // hands the context back when the call ends, whichever way
  SmallContext* c = nullptr;
  SmallPool* pool = nullptr;
  bool at_rest = false;   // the call has waited for its stream: nothing queued on it still touches the context's buffers
  ~SmallContextLease() {
    if (!c) return;
    // An early return (a failed copy or launch) may leave a copy or the kernel in flight on the pinned chunks and the
    // device buffer; the next borrower would repack them underneath that work.
    if (!at_rest) {
      (void)hipStreamSynchronize(c->stream);
      (void)hipGetLastError();
    }
    std::lock_guard<std::mutex> lock(pool->mu);
    pool->idle.push_back(c);
  }
};
int gemm_host_small(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k, const double alpha[2], const void* a,
                    int64_t lda, const void* b, int64_t ldb, const double beta[2], void* c, int64_t ldc) {
  const int es = kElemSize[dtype];
  const int64_t a_rows = stored_rows_is_m(opa) ? m : k, a_cols = stored_rows_is_m(opa) ? k : m;
  const int64_t b_rows = stored_rows_is_m(opb) ? k : n, b_cols = stored_rows_is_m(opb) ? n : k;
  const int64_t dlda = round_up(std::max<int64_t>(a_rows, 1), 16), dldb = round_up(std::max<int64_t>(b_rows, 1), 16);
  const int64_t dldc = round_up(m, 16);
  const bool beta_is_zero = beta[0] == 0.0 && beta[1] == 0.0;
  const size_t a_bytes = k > 0 ? (size_t)dlda * a_cols * es : 0, b_bytes = k > 0 ? (size_t)dldb * b_cols * es : 0;
  const size_t c_bytes = (size_t)dldc * n * es;
  const size_t off_b = round_up((int64_t)a_bytes, 256), off_c = off_b + round_up((int64_t)b_bytes, 256);
  const size_t up_bytes = beta_is_zero ? off_c : off_c + c_bytes;

  if (off_c + c_bytes > kSmallContextBytes) return fail(EIGAMD_ERR_ARG, "small host-pointer product beyond its context");
  SmallContextLease lease;
  lease.pool = &small_pool();
  int rc = small_context_acquire(&lease.c);
  if (rc != EIGAMD_OK) return rc;
  SmallContext& st = *lease.c;
  char* pin = st.up;
  auto pack = [&](char* dst, int64_t dld, const void* src, int64_t ld, int64_t rows, int64_t cols) {
    const char* sp = static_cast<const char*>(src);
    if (ld == dld && cols > 0) { std::memcpy(dst, sp, ((size_t)ld * (cols - 1) + rows) * es); return; }  // (the last column ends after `rows`)
    for (int64_t j = 0; j < cols; ++j) std::memcpy(dst + (size_t)j * dld * es, sp + (size_t)j * ld * es, (size_t)rows * es);
  };
  if (k > 0) {
    pack(pin, dlda, a, lda, a_rows, a_cols);
    pack(pin + off_b, dldb, b, ldb, b_rows, b_cols);
  }
  if (!beta_is_zero) pack(pin + off_c, dldc, c, ldc, m, n);
  // The smallest products skip both copies: the kernel reads its operands from the pinned chunk over the bus and
  // writes C back into it (pinned host memory is addressable from the device); up to about a megabyte of operands
  // two DMA transfers cost more than that (dgemm_ 64^3 43 -> 31 us, 128^3 58 -> 45, 192^3 93 -> 75; even at 256^3,
  // slower from 320^3).  EIGAMD_HOST_ZERO_COPY_BYTES moves the limit (0: always copy).
  static const size_t zero_copy_limit = [] {
    const char* e = getenv("EIGAMD_HOST_ZERO_COPY_BYTES");
    return e ? (size_t)atoll(e) : (size_t)1 << 20;
  }();
  const bool zero_copy = off_c + c_bytes <= zero_copy_limit;
  char* dev = zero_copy ? pin : st.dev;
  if (!zero_copy && up_bytes > 0) HIP_TRY(hipMemcpyAsync(dev, pin, up_bytes, hipMemcpyHostToDevice, st.stream));
  GemmCall call;
  call.kind = dtype; call.opa = opa; call.opb = opb;
  call.m = m; call.n = n; call.k = k;
  call.alpha[0] = alpha[0]; call.alpha[1] = alpha[1];
  call.beta[0] = beta[0]; call.beta[1] = beta[1];
  call.A = dev; call.lda = dlda;
  call.B = dev + off_b; call.ldb = dldb;
  call.C = dev + off_c; call.ldc = dldc;
  call.serial_edges = true;  // (a product this small is one launch anyway; no helper streams)
  HIP_TRY(launch(call, st.stream));
  char* back = zero_copy ? pin + off_c : st.down;
  if (!zero_copy) HIP_TRY(hipMemcpyAsync(back, dev + off_c, c_bytes, hipMemcpyDeviceToHost, st.stream));
  HIP_TRY(hipStreamSynchronize(st.stream));
  lease.at_rest = true;
  // only the m x n window goes back: rows m..ldc-1 of C keep the caller's bytes
  char* cp = static_cast<char*>(c);
  for (int64_t j = 0; j < n; ++j) std::memcpy(cp + (size_t)j * ldc * es, back + (size_t)j * dldc * es, (size_t)m * es);
  return EIGAMD_OK;
}

int gemm_host_impl(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k,
                   const double alpha[2], const void* a, int64_t lda, const void* b, int64_t ldb,
                   const double beta[2], void* c, int64_t ldc, int64_t c_inc = 1) {
  if (m == 0 || n == 0) return EIGAMD_OK;                          // level3_impl.h:59-60
  const bool beta_is_one = beta[0] == 1.0 && beta[1] == 0.0;
  if (k == 0 && beta_is_one) return EIGAMD_OK;                     // level3_impl.h:62-69
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;

  // On request (multi_gpu_rule) a large product is dealt out over all visible GPUs (eigamd_pgemm_host), as
  // the reference splits one over the cores without the caller noticing (parallelize_gemm,
  // Eigen/src/Core/products/Parallelizer.h:101-176).
  if (c_inc == 1 && k > 0) {
    MultiGpuRule rule = multi_gpu_rule();
    // Panels of the reduction: the products of the last one run with nothing left to upload beside them, so there
    // should be at least eight (a 1024-deep local product still runs at nine tenths of a 2048-deep one).
    if (!getenv("EIGAMD_MULTI_GPU_NB") && k < 8 * rule.nb) rule.nb = 1024;
    if (rule.ranks > 1 && (double)m * (double)n * (double)k >= rule.min_work && m >= 2 * rule.nb && n >= 2 * rule.nb) {
      int nprow = 1;
      for (int r = 1; r * r <= rule.ranks; ++r)
        if (rule.ranks % r == 0) nprow = r;
      std::vector<int> devs(rule.ranks);
      int cur = 0;
      (void)hipGetDevice(&cur);
      for (int i = 0; i < rule.ranks; ++i) devs[i] = rule.virtual_grid ? cur : i;
      double al[2] = {alpha[0], alpha[1]}, be[2] = {beta[0], beta[1]};
      float alf[2] = {(float)alpha[0], (float)alpha[1]}, bef[2] = {(float)beta[0], (float)beta[1]};
      const bool single = dtype == F32 || dtype == C32;
      return eigamd_pgemm_host(dtype, opa, opb, m, n, k, single ? (const void*)alf : (const void*)al, a, lda, b, ldb,
                               single ? (const void*)bef : (const void*)be, c, ldc, nprow, rule.ranks / nprow, rule.nb, devs.data(), 0);
    }
  }

  const int es = kElemSize[dtype];
  const int64_t a_rows = stored_rows_is_m(opa) ? m : k, a_cols = stored_rows_is_m(opa) ? k : m;
  {
    const int64_t b_rows = stored_rows_is_m(opb) ? k : n, b_cols = stored_rows_is_m(opb) ? n : k;
    const size_t total = ((size_t)round_up(a_rows, 16) * a_cols + (size_t)round_up(b_rows, 16) * b_cols + (size_t)round_up(m, 16) * n) * es;
    if (c_inc == 1 && total <= kSmallProductBytes)
      return gemm_host_small(dtype, opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
  }
  // The device images are this library's own, so their m and n extents are rounded up to whole
  // tiles and the kernel is launched on the rounded extents: every workgroup then runs the
  // vector-load instantiation and no bounds-checked edge launch is needed.  What the padding
  // holds does not matter: a row of op(A) beyond m only feeds rows of C beyond m, a column of
  // op(B) beyond n only columns beyond n, and only the m x n window of C is copied back.  (k is
  // not padded: the kernel zero-fills its last partial k-tile itself.)
  int bm = 1, bn = 1, bk = 1;
  eigamd_tile_shape(dtype, &bm, &bn, &bk);
  const int64_t mp = round_up(m, bm), np = round_up(n, bn);
  const int64_t a_rows_p = stored_rows_is_m(opa) ? mp : k, a_cols_p = stored_rows_is_m(opa) ? k : mp;
  const int64_t b_rows_p = stored_rows_is_m(opb) ? k : np, b_cols_p = stored_rows_is_m(opb) ? np : k;
  // Device leading dimensions: 16-byte aligned rows so the vector kernels apply.
  const int64_t dlda = round_up(std::max<int64_t>(a_rows_p, 1), 16);
  const int64_t dldb = round_up(std::max<int64_t>(b_rows_p, 1), 16);
  const int64_t dldc = round_up(mp, 16);
  const bool beta_is_zero = beta[0] == 0.0 && beta[1] == 0.0;

  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  if ((rc = st.init()) != EIGAMD_OK) return rc;
  if (k > 0) {
    if ((rc = st.reserve(0, (size_t)dlda * a_cols_p * es)) != EIGAMD_OK) return rc;
    if ((rc = st.reserve(1, (size_t)dldb * b_cols_p * es)) != EIGAMD_OK) return rc;
  }
  if ((rc = st.reserve(2, (size_t)dldc * np * es)) != EIGAMD_OK) return rc;

  // Pipeline over column panels of C: while the kernel works on panel j, the host threads
  // stage panel j+1 (B and, unless beta == 0, C) and bring panel j-1 back.  A goes first.
  if (k > 0 && (rc = st.upload(st.buf[0], dlda, a, lda, a_rows, a_cols, es)) != EIGAMD_OK) return rc;
  int64_t panel = round_up(std::max<int64_t>(1, (int64_t)(256u << 20) / std::max<int64_t>(1, m * es)), 256);
  panel = std::min(panel, n);
  hipEvent_t computed[2] = {st.computed, st.computed2};
  auto cpanel = [&](int64_t j0) { return static_cast<char*>(st.buf[2]) + (size_t)j0 * dldc * es; };
  auto fetch = [&](int64_t j0, int slot) -> int {  // panel [j0, j0 + nb) of C back to the caller
    const int64_t nb = std::min(panel, n - j0);
    HIP_TRY(hipStreamWaitEvent(st.d2h, computed[slot], 0));
    // Only the m x n window goes back: rows m..ldc-1 of C keep the caller's bytes.
    return st.download(static_cast<char*>(c) + (size_t)j0 * ldc * es, ldc, cpanel(j0), dldc, m, nb, es, c_inc);
  };
  int slot = 0;
  int64_t prev = -1;
  for (int64_t j0 = 0; j0 < n; j0 += panel, slot ^= 1) {
    const int64_t nb = std::min(panel, n - j0);
    const char* bdev = static_cast<const char*>(st.buf[1]);
    if (k > 0) {
      if (stored_rows_is_m(opb))  // B is k x n: columns j0..
        rc = st.upload(const_cast<char*>(bdev) + (size_t)j0 * dldb * es, dldb,
                       static_cast<const char*>(b) + (size_t)j0 * ldb * es, ldb, k, nb, es);
      else                        // B is n x k: rows j0..
        rc = st.upload(const_cast<char*>(bdev) + (size_t)j0 * es, dldb, static_cast<const char*>(b) + (size_t)j0 * es, ldb,
                       nb, k, es);
      if (rc != EIGAMD_OK) return rc;
    }
    if (!beta_is_zero &&
        (rc = st.upload(cpanel(j0), dldc, static_cast<const char*>(c) + (size_t)j0 * ldc * es, ldc, m, nb, es, c_inc)) != EIGAMD_OK)
      return rc;
    HIP_TRY(hipEventRecord(st.uploaded, st.h2d));
    HIP_TRY(hipStreamWaitEvent(st.stream, st.uploaded, 0));
    GemmCall call;
    call.kind = dtype; call.opa = opa; call.opb = opb;
    call.m = mp; call.n = round_up(nb, bn); call.k = k;   // whole tiles (the panel width is a multiple of every BN)
    call.alpha[0] = alpha[0]; call.alpha[1] = alpha[1];
    call.beta[0] = beta[0]; call.beta[1] = beta[1];
    call.A = st.buf[0]; call.lda = dlda;
    call.B = stored_rows_is_m(opb) ? bdev + (size_t)j0 * dldb * es : bdev + (size_t)j0 * es;
    call.ldb = dldb;
    call.C = cpanel(j0); call.ldc = dldc;
    HIP_TRY(launch(call, st.stream));
    HIP_TRY(hipEventRecord(computed[slot], st.stream));
    if (prev >= 0 && (rc = fetch(prev, slot ^ 1)) != EIGAMD_OK) return rc;
    prev = j0;
  }
  return fetch(prev, slot ^ 1);
}

int op_of(char ch) {  // blas/common.h:47-50
  switch (ch) {
    case 'N': case 'n': return OP_N;
    case 'T': case 't': return OP_T;
    case 'C': case 'c': return OP_C;
  }
  return -1;
}

}  // namespace

namespace eigamd {
int abi_fail(int code, const std::string& what) { return fail(code, what); }  // for the other translation units of the ABI

// rows x cols elements between a caller's (pageable) matrix and memory of the current device, through
// this device's pinned chunks and copy threads; returns when the data has arrived (with wait = false: once it is
// queued; staged_upload_wait then waits for every upload and brings every queued download home).  (Asynchronous 2-D
// copies straight from pageable memory, issued on several streams for neighbouring blocks of one host
// matrix, made the runtime abort now and then; the staged form is also five times faster.)
int staged_copy(bool to_device, void* dev, int64_t dld, void* host, int64_t ld, int64_t rows, int64_t cols, int es, bool wait) {
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  int rc = st.init();
  if (rc != EIGAMD_OK) return rc;
  if (to_device) {
    if ((rc = st.upload(dev, dld, host, ld, rows, cols, es)) != EIGAMD_OK) return rc;
    if (wait) HIP_TRY(hipStreamSynchronize(st.h2d));
    return EIGAMD_OK;
  }
  if (!wait) return st.download_queue(host, ld, dev, dld, rows, cols, es);
  return st.download(host, ld, dev, dld, rows, cols, es);
}
int staged_upload_wait() {
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  int rc = st.init();
  if (rc != EIGAMD_OK) return rc;
  HIP_TRY(hipStreamSynchronize(st.h2d));
  return st.download_flush();
}
namespace {
struct EdgeStreams {
  std::mutex mu;
  hipStream_t s[2] = {nullptr, nullptr};
  hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
  bool off = false;
};
EdgeStreams& edge_streams() {
  static EdgeStreams e[kMaxDevices];
  return e[current_device_slot()];
}
}  // namespace

hipError_t edge_streams_fork(hipStream_t s, hipStream_t* a, hipStream_t* b) {
  EdgeStreams& es = edge_streams();
  std::lock_guard<std::mutex> lock(es.mu);
  *a = *b = s;
  if (es.off) return hipSuccess;
  if (!es.s[0]) {
    if (const char* e = getenv("EIGAMD_EDGE_STREAMS")) es.off = atoi(e) == 0;
    if (es.off) return hipSuccess;
    for (int i = 0; i < 2; ++i) {
      hipError_t e = hipStreamCreateWithFlags(&es.s[i], hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&es.join[i], hipEventDisableTiming);
      if (e != hipSuccess) return e;
    }
    hipError_t e = hipEventCreateWithFlags(&es.fork, hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  // An event's record is taken at enqueue time: calls from several host threads may share the
  // events as long as each record / wait pair is issued under the lock.
  hipError_t e = hipEventRecord(es.fork, s);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipStreamWaitEvent(es.s[i], es.fork, 0);
  if (e != hipSuccess) return e;
  *a = es.s[0];
  *b = es.s[1];
  return hipSuccess;
}

namespace {
// A small pool of buffers per device for the partial products of split reductions.  Each buffer remembers the stream
// that used it last.  While every stream that asks finds a buffer of its own (at most kSlots streams), a buffer never
// changes hands and the stream's own order is all the ordering there is to keep: nothing is recorded, nothing waited
// for (an event recorded behind every use costs the stream a marker, 5 to 6 us between two kernels -- a tenth of a
// 1024^3 product).  A stream is told from a later one that happens to get the same handle by its id.  The first time a
// buffer has to change hands the device is brought to rest once, and from then on every use is fenced by an event
// (recorded at the hand-back, waited for by the next taker), which is always right and was the only mode before.
struct SplitWorkspaces {
  static constexpr int kSlots = 8;
  std::mutex mu;
  int off = -1;  // -1: environment not read yet
  bool fenced = false;   // every use is followed by an event (see above)
  uint64_t clock = 0;
  struct Slot {
    hipStream_t stream = nullptr;
    unsigned long long stream_id = 0;
    void* buf = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;   // recorded by splitk_release (fenced mode)
    bool released = true, used = false;
    uint64_t stamp = 0;
  } slot[kSlots];
};
SplitWorkspaces& split_workspaces() {
  static SplitWorkspaces w[kMaxDevices];
  return w[current_device_slot()];
}
}  // namespace

void* splitk_acquire(hipStream_t s, size_t bytes, int* slot_out) {
  *slot_out = -1;
  if (s == hipStreamPerThread) return nullptr;  // one handle value, a different stream in every host thread
  SplitWorkspaces& w = split_workspaces();
  std::lock_guard<std::mutex> lock(w.mu);
  if (w.off < 0) {
    const char* e = getenv("EIGAMD_SPLITK");
    w.off = (e && atoi(e) == 0) ? 1 : 0;
    if (const char* f = getenv("EIGAMD_SPLITK_FENCED")) w.fenced = atoi(f) != 0;   // (measurements: the events from the start)
  }
  if (w.off) return nullptr;
  unsigned long long id = 0;
  // (hipStreamGetId is looked up at run time: it is younger than the runtime some hosts bring along -- a PyTorch wheel
  // ships its own libamdhip64 -- and a library that required it would not load there.  Where it is missing: the events.)
  typedef hipError_t (*StreamIdFn)(hipStream_t, unsigned long long*);
  static const StreamIdFn stream_id = reinterpret_cast<StreamIdFn>(dlsym(RTLD_DEFAULT, "hipStreamGetId"));
  if (!w.fenced && (!stream_id || stream_id(s, &id) != hipSuccess)) {   // no way to tell streams apart: the events, then
    (void)hipGetLastError();
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    w.fenced = true;
  }
  int pick = -1;
  if (!w.fenced) {
    // this stream's own buffer, else one nobody has used yet
    for (int i = 0; i < SplitWorkspaces::kSlots && pick < 0; ++i)
      if (w.slot[i].used && w.slot[i].released && w.slot[i].stream == s && w.slot[i].stream_id == id) pick = i;
    for (int i = 0; i < SplitWorkspaces::kSlots && pick < 0; ++i)
      if (!w.slot[i].used) pick = i;
    if (pick < 0) {
      // A buffer has to change hands (more streams than buffers, or this stream's buffer is in a caller's hands):
      // whatever the buffers' last users queued is allowed to finish, once, and the events take over.
      if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      w.fenced = true;
      for (int i = 0; i < SplitWorkspaces::kSlots; ++i) w.slot[i].used = w.slot[i].used && !w.slot[i].released;   // (at rest: no last use to wait for)
    }
  }
  if (w.fenced) {
    // the buffer this stream used last; else the least recently used one that is not in the hands of
    // a caller right now (between acquire and release)
    for (int i = 0; i < SplitWorkspaces::kSlots && pick < 0; ++i)
      if (w.slot[i].used && w.slot[i].stream == s && w.slot[i].released) pick = i;
    if (pick < 0)
      for (int i = 0; i < SplitWorkspaces::kSlots; ++i)
        if (w.slot[i].released && (pick < 0 || w.slot[i].stamp < w.slot[pick].stamp)) pick = i;
  }
  if (pick < 0) return nullptr;
  SplitWorkspaces::Slot& t = w.slot[pick];
  if (w.fenced) {
    // Behind the buffer's last use.  (Also when the handle is the same: a destroyed stream's handle
    // value can come back as a new stream.)
    if (t.used && t.done && hipStreamWaitEvent(s, t.done, 0) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    if (!t.done && hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      t.done = nullptr;
      return nullptr;
    }
  }
  if (t.bytes < bytes) {
    // hipFree waits for the device, so nothing queued earlier still reads the old buffer.
    if (t.buf) (void)hipFree(t.buf);
    t.buf = nullptr; t.bytes = 0;
    const size_t want = std::max(bytes, (size_t)32 << 20);
    if (hipMalloc(&t.buf, want) != hipSuccess) {
      (void)hipGetLastError();
      t.buf = nullptr;
      return nullptr;
    }
    t.bytes = want;
  }
  t.stream = s; t.stream_id = id; t.used = true; t.released = false; t.stamp = ++w.clock;
  *slot_out = pick;
  return t.buf;
}

void splitk_release(int slot, hipStream_t s) {
  SplitWorkspaces& w = split_workspaces();
  std::lock_guard<std::mutex> lock(w.mu);
  if (slot < 0 || slot >= SplitWorkspaces::kSlots) return;
  SplitWorkspaces::Slot& t = w.slot[slot];
  if (w.fenced) {
    if (!t.done && hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      t.done = nullptr;
      (void)hipStreamSynchronize(s);   // no event to be had: the use is over before anybody else can take the buffer
    }
    if (t.done) (void)hipEventRecord(t.done, s);
  }
  t.released = true;
}

hipError_t edge_streams_join(hipStream_t s, hipStream_t a, hipStream_t b) {
  if (a == s && b == s) return hipSuccess;
  EdgeStreams& es = edge_streams();
  std::lock_guard<std::mutex> lock(es.mu);
  hipError_t e = hipEventRecord(es.join[0], a);
  if (e == hipSuccess) e = hipStreamWaitEvent(s, es.join[0], 0);
  if (e == hipSuccess) e = hipEventRecord(es.join[1], b);
  if (e == hipSuccess) e = hipStreamWaitEvent(s, es.join[1], 0);
  return e;
}
}  // namespace eigamd

extern "C" {

// The process's xerbla_ wins when one is linked in (the Netlib testers install
// their own); this weak default prints like blas/xerbla.cpp:15-19.
__attribute__((weak)) int xerbla_(const char* msg, int* info, int) {
  std::printf("Eigen BLAS ERROR #%i: %s\n", *info, msg);
  return 0;
}

static int f77_gemm(int dtype, const char* name, const char* transa, const char* transb,
                    const int* m, const int* n, const int* k, const void* alpha, const void* a,
                    const int* lda, const void* b, const int* ldb, const void* beta, void* c,
                    const int* ldc) {
  const int opa = op_of(*transa), opb = op_of(*transb);
  int info = 0;
  if (opa < 0) info = 1;
  else if (opb < 0) info = 2;
  else info = check_args(opa, opb, *m, *n, *k, *lda, *ldb, *ldc);
  if (info) return xerbla_(name, &info, 6);
  double al[2], be[2];
  read_scalar(dtype, alpha, al);
  read_scalar(dtype, beta, be);
  int rc = gemm_host_impl(dtype, opa, opb, *m, *n, *k, al, a, *lda, b, *ldb, be, c, *ldc);
  if (rc != EIGAMD_OK) {
    // The F77 ABI has no error channel besides xerbla_; a device failure must not
    // pass silently as a computed product.
    std::fprintf(stderr, "eigen_amd %s: %s\n", name, g_err.c_str());
    std::abort();
  }
  return 0;
}

int sgemm_(const char* ta, const char* tb, const int* m, const int* n, const int* k,
           const float* alpha, const float* a, const int* lda, const float* b, const int* ldb,
           const float* beta, float* c, const int* ldc) {
  return f77_gemm(F32, "SGEMM ", ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}
int dgemm_(const char* ta, const char* tb, const int* m, const int* n, const int* k,
           const double* alpha, const double* a, const int* lda, const double* b, const int* ldb,
           const double* beta, double* c, const int* ldc) {
  return f77_gemm(F64, "DGEMM ", ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}
int cgemm_(const char* ta, const char* tb, const int* m, const int* n, const int* k,
           const float* alpha, const float* a, const int* lda, const float* b, const int* ldb,
           const float* beta, float* c, const int* ldc) {
  return f77_gemm(C32, "CGEMM ", ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}
int zgemm_(const char* ta, const char* tb, const int* m, const int* n, const int* k,
           const double* alpha, const double* a, const int* lda, const double* b, const int* ldb,
           const double* beta, double* c, const int* ldc) {
  return f77_gemm(C64, "ZGEMM ", ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}

int eigamd_gemm_host(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k,
                     const void* alpha, const void* a, int64_t lda, const void* b, int64_t ldb,
                     const void* beta, void* c, int64_t ldc) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  const int info = check_args(opa, opb, m, n, k, lda, ldb, ldc);
  if (info) return fail(EIGAMD_ERR_ARG, "invalid argument #" + std::to_string(info));
  if (m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "extent exceeds 2^31-1");
  double al[2], be[2];
  read_scalar(dtype, alpha, al);
  read_scalar(dtype, beta, be);
  return gemm_host_impl(dtype, opa, opb, m, n, k, al, a, lda, b, ldb, be, c, ldc);
}

// Staging for a product with one real operand: everything goes up, one device call, the
// m x n window comes back (no column-panel pipeline: these forms are the rare ones).
static int gemm_mixed_host_impl(int dtype, bool a_real, bool b_real, int opa, int opb, int64_t m, int64_t n, int64_t k,
                                const double alpha[2], const void* a, int64_t lda, const void* b, int64_t ldb,
                                const double beta[2], void* c, int64_t ldc, int64_t c_inc) {
  if (m == 0 || n == 0) return EIGAMD_OK;
  const bool beta_is_one = beta[0] == 1.0 && beta[1] == 0.0, beta_is_zero = beta[0] == 0.0 && beta[1] == 0.0;
  if (k == 0 && beta_is_one) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const int es = kElemSize[dtype], esa = a_real ? es / 2 : es, esb = b_real ? es / 2 : es;
  const int64_t a_rows = stored_rows_is_m(opa) ? m : k, a_cols = stored_rows_is_m(opa) ? k : m;
  const int64_t b_rows = stored_rows_is_m(opb) ? k : n, b_cols = stored_rows_is_m(opb) ? n : k;
  const int64_t dlda = round_up(std::max<int64_t>(a_rows, 1), 16), dldb = round_up(std::max<int64_t>(b_rows, 1), 16);
  const int64_t dldc = round_up(m, 16);
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  if ((rc = st.init()) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(0, (size_t)dlda * std::max<int64_t>(a_cols, 1) * esa)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(1, (size_t)dldb * std::max<int64_t>(b_cols, 1) * esb)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(2, (size_t)dldc * n * es)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[0], dlda, a, lda, a_rows, a_cols, esa)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[1], dldb, b, ldb, b_rows, b_cols, esb)) != EIGAMD_OK) return rc;
  if (!beta_is_zero && (rc = st.upload(st.buf[2], dldc, c, ldc, m, n, es, c_inc)) != EIGAMD_OK) return rc;
  HIP_TRY(hipEventRecord(st.uploaded, st.h2d));
  HIP_TRY(hipStreamWaitEvent(st.stream, st.uploaded, 0));
  HIP_TRY(gemm_mixed_device(dtype, a_real, b_real, opa, opb, m, n, k, alpha, st.buf[0], dlda, st.buf[1], dldb, beta,
                            st.buf[2], dldc, st.stream));
  HIP_TRY(hipEventRecord(st.computed, st.stream));
  HIP_TRY(hipStreamWaitEvent(st.d2h, st.computed, 0));
  return st.download(c, ldc, st.buf[2], dldc, m, n, es, c_inc);
}

static int check_ex(int dtype, int a_real, int b_real, int opa, int opb, int64_t m, int64_t n, int64_t k, int64_t lda,
                    int64_t ldb, int64_t c_inc, int64_t ldc) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  if ((a_real || b_real) && (dtype < C32 || (a_real && b_real)))
    return fail(EIGAMD_ERR_ARG, "a real operand needs a complex result kind and a complex other operand");
  if (c_inc < 1) return fail(EIGAMD_ERR_ARG, "inner stride of C must be positive");
  int info = check_args(opa, opb, m, n, k, lda, ldb, std::max<int64_t>(ldc, std::max<int64_t>(1, m)));
  // ldc >= max(1, m) for a dense C (blas/level3_impl.h:54); the columns of a strided C must not overlap
  if (!info && ldc < (m > 0 ? (m - 1) * c_inc + 1 : 1)) info = 13;
  if (info) return fail(EIGAMD_ERR_ARG, "invalid argument #" + std::to_string(info));
  if (m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff) return fail(EIGAMD_ERR_ARG, "extent exceeds 2^31-1");
  return EIGAMD_OK;
}

int eigamd_gemm_ex_host(int dtype, int a_is_real, int b_is_real, int opa, int opb, int64_t m, int64_t n, int64_t k,
                        const void* alpha, const void* a, int64_t lda, const void* b, int64_t ldb, const void* beta,
                        void* c, int64_t c_inc, int64_t ldc) {
  int rc = check_ex(dtype, a_is_real, b_is_real, opa, opb, m, n, k, lda, ldb, c_inc, ldc);
  if (rc != EIGAMD_OK) return rc;
  double al[2], be[2];
  read_scalar(dtype, alpha, al);
  read_scalar(dtype, beta, be);
  if (!a_is_real && !b_is_real) return gemm_host_impl(dtype, opa, opb, m, n, k, al, a, lda, b, ldb, be, c, ldc, c_inc);
  return gemm_mixed_host_impl(dtype, a_is_real != 0, b_is_real != 0, opa, opb, m, n, k, al, a, lda, b, ldb, be, c, ldc, c_inc);
}

int eigamd_gemm_mixed_dev(int dtype, int a_is_real, int b_is_real, int opa, int opb, int64_t m, int64_t n, int64_t k,
                          const void* alpha, const void* d_a, int64_t lda, const void* d_b, int64_t ldb, const void* beta,
                          void* d_c, int64_t ldc, void* stream) {
  if (!a_is_real && !b_is_real) return fail(EIGAMD_ERR_ARG, "no real operand: use eigamd_gemm_dev");
  int rc = check_ex(dtype, a_is_real, b_is_real, opa, opb, m, n, k, lda, ldb, 1, ldc);
  if (rc != EIGAMD_OK) return rc;
  if (m == 0 || n == 0) return EIGAMD_OK;
  if ((rc = ensure_device()) != EIGAMD_OK) return rc;
  double al[2], be[2];
  read_scalar(dtype, alpha, al);
  read_scalar(dtype, beta, be);
  if (k == 0 && be[0] == 1.0 && be[1] == 0.0) return EIGAMD_OK;
  HIP_TRY(gemm_mixed_device(dtype, a_is_real != 0, b_is_real != 0, opa, opb, m, n, k, al, d_a, lda, d_b, ldb, be, d_c, ldc,
                            static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

int eigamd_gemm_dev(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k,
                    const void* alpha, const void* d_a, int64_t lda, const void* d_b,
                    int64_t ldb, const void* beta, void* d_c, int64_t ldc, void* stream) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  const int info = check_args(opa, opb, m, n, k, lda, ldb, ldc);
  if (info) return fail(EIGAMD_ERR_ARG, "invalid argument #" + std::to_string(info));
  if (m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "extent exceeds 2^31-1");
  if (m == 0 || n == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  GemmCall call;
  call.kind = dtype; call.opa = opa; call.opb = opb;
  call.m = m; call.n = n; call.k = k;
  read_scalar(dtype, alpha, call.alpha);
  read_scalar(dtype, beta, call.beta);
  if (k == 0 && call.beta[0] == 1.0 && call.beta[1] == 0.0) return EIGAMD_OK;
  call.A = d_a; call.lda = lda;
  call.B = d_b; call.ldb = ldb;
  call.C = d_c; call.ldc = ldc;
  // EIGAMD_GEMM_ATOMIC=1 (experiments): beta == 1 products of the real kinds add into C at the memory side
  static const bool atomic = [] { const char* e = getenv("EIGAMD_GEMM_ATOMIC"); return e && atoi(e) != 0; }();
  call.atomic_c = atomic;
  HIP_TRY(launch(call, static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

}  // extern "C"

// ---- the entry point that allocates nothing ---------------------------------------------
//
// eigamd_gemm_dev may call hipMalloc on its way (the partial-product buffer of a split
// reduction, the helper streams of the edge launches).  Callers that must not allocate get the
// same product with the partial products in a fixed device array that is part of the library's
// code object: every GPU the library runs on holds its own copy from load to exit.
namespace {
constexpr size_t kNoAllocScratchBytes = (size_t)128 << 20;
constexpr size_t kNoAllocMaxCBytes = kNoAllocScratchBytes / 16;  // up to 16 pieces of the reduction (choose_splits) fit
}
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];
namespace {
// The array's name is a host-side stand-in, not a device address: the address on each device
// is asked for once, with that device current, and only that pointer is ever handed to a kernel.
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};
int noalloc_scratch(void** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(EIGAMD_ERR_DEVICE, "device ordinal beyond the library's table");
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(EIGAMD_ERR_DEVICE, "hipGetSymbolAddress returned no address for the scratch array");
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return EIGAMD_OK;
}
}  // namespace

extern "C" {

int eigamd_gemm_dev_noalloc(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k,
                            const void* alpha, const void* d_a, int64_t lda, const void* d_b,
                            int64_t ldb, const void* beta, void* d_c, int64_t ldc, void* stream) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  const int info = check_args(opa, opb, m, n, k, lda, ldb, ldc);
  if (info) return fail(EIGAMD_ERR_ARG, "invalid argument #" + std::to_string(info));
  if (m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "extent exceeds 2^31-1");
  if ((uint64_t)m * (uint64_t)n * kElemSize[dtype] > kNoAllocMaxCBytes)
    return fail(EIGAMD_ERR_ARG, "eigamd_gemm_dev_noalloc: m * n * sizeof(scalar) = " +
                                    std::to_string((uint64_t)m * (uint64_t)n * kElemSize[dtype]) + " bytes exceeds the maximum of " +
                                    std::to_string(kNoAllocMaxCBytes) + " (use eigamd_gemm_dev)");
  if (m == 0 || n == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  GemmCall call;
  call.kind = dtype; call.opa = opa; call.opb = opb;
  call.m = m; call.n = n; call.k = k;
  read_scalar(dtype, alpha, call.alpha);
  read_scalar(dtype, beta, call.beta);
  if (k == 0 && call.beta[0] == 1.0 && call.beta[1] == 0.0) return EIGAMD_OK;
  call.A = d_a; call.lda = lda;
  call.B = d_b; call.ldb = ldb;
  call.C = d_c; call.ldc = ldc;
  void* area = nullptr;
  if ((rc = noalloc_scratch(&area)) != EIGAMD_OK) return rc;
  // A product this small is at most a few rounds of tiles: one launch over everything (no edge
  // launches, hence no helper streams), its reduction split where that pays, the pieces in `area`.
  // Every piece is written in full before the pass that adds them up reads it, so nothing in
  // the array has to start at zero; two calls whose work overlaps on one GPU would share it.
  call.partials = area;
  call.partials_bytes = kNoAllocScratchBytes;
  HIP_TRY(launch(call, static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

int eigamd_fill_uniform_dev(int dtype, void* d_x, int64_t rows, int64_t cols, int64_t ld,
                            uint64_t seed, void* stream) {
  if (dtype < 0 || dtype > 3 || rows < 0 || cols < 0 || ld < std::max<int64_t>(rows, 1))
    return fail(EIGAMD_ERR_ARG, "invalid fill arguments");
  if (rows == 0 || cols == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const CyclicMap ident = {1, 1, 0};
  HIP_TRY(fill_uniform_device(dtype, d_x, rows, cols, ld, seed, ident, ident, static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

int eigamd_fill_uniform_cyclic_dev(int dtype, void* d_x, int64_t rows, int64_t cols, int64_t ld,
                                   uint64_t seed, int64_t nb_r, int nprow, int myrow, int64_t nb_c,
                                   int npcol, int mycol, void* stream) {
  if (dtype < 0 || dtype > 3 || rows < 0 || cols < 0 || ld < std::max<int64_t>(rows, 1) || nb_r < 1 ||
      nb_c < 1 || nprow < 1 || npcol < 1 || myrow < 0 || myrow >= nprow || mycol < 0 || mycol >= npcol)
    return fail(EIGAMD_ERR_ARG, "invalid fill arguments");
  if (rows == 0 || cols == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const CyclicMap rmap = {nb_r, nprow, myrow}, cmap = {nb_c, npcol, mycol};
  HIP_TRY(fill_uniform_device(dtype, d_x, rows, cols, ld, seed, rmap, cmap, static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

int eigamd_fill_uniform_host(int dtype, void* x, int64_t rows, int64_t cols, int64_t ld,
                             uint64_t seed) {
  if (dtype < 0 || dtype > 3 || rows < 0 || cols < 0 || ld < std::max<int64_t>(rows, 1))
    return fail(EIGAMD_ERR_ARG, "invalid fill arguments");
  fill_uniform_host(dtype, x, rows, cols, ld, seed);
  return EIGAMD_OK;
}

namespace {
int getrf_dev_impl(int dtype, int64_t m, int64_t n, void* d_a, int64_t lda, int* d_transp, int64_t max_block_size,
                   int* first_zero_pivot, int* nb_transpositions, void* stream, bool allow_fused) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "invalid dtype");
  if (m < 0 || n < 0 || lda < std::max<int64_t>(1, m) || m > 0x7fffffff || n > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid LU arguments");
  int fz = -1, nb = 0;
  if (m > 0 && n > 0) {
    int rc = ensure_device();
    if (rc != EIGAMD_OK) return rc;
    const hipError_t e =
        getrf_device(dtype, m, n, d_a, lda, d_transp, max_block_size, &fz, &nb, static_cast<hipStream_t>(stream), allow_fused);
    if (e == kLuNotResident)
      return fail(EIGAMD_ERR_NOT_RESIDENT,
                  "LU: the workgroups of a panel kernel were not resident together (device shared with other work); "
                  "the matrix is partly overwritten: restore it and call eigamd_getrf_dev_shared");
    if (e != hipSuccess) return hip_fail(e, "getrf_device");
  }
  if (first_zero_pivot) *first_zero_pivot = fz;
  if (nb_transpositions) *nb_transpositions = nb;
  return EIGAMD_OK;
}
}  // namespace

int eigamd_getrf_dev(int dtype, int64_t m, int64_t n, void* d_a, int64_t lda, int* d_transp,
                     int64_t max_block_size, int* first_zero_pivot, int* nb_transpositions, void* stream) {
  return getrf_dev_impl(dtype, m, n, d_a, lda, d_transp, max_block_size, first_zero_pivot, nb_transpositions, stream, true);
}

int eigamd_getrf_dev_shared(int dtype, int64_t m, int64_t n, void* d_a, int64_t lda, int* d_transp,
                            int64_t max_block_size, int* first_zero_pivot, int* nb_transpositions, void* stream) {
  return getrf_dev_impl(dtype, m, n, d_a, lda, d_transp, max_block_size, first_zero_pivot, nb_transpositions, stream, false);
}

int eigamd_getrf_dev_async(int dtype, int64_t m, int64_t n, void* d_a, int64_t lda, int* d_transp, int64_t max_block_size,
                           int* d_info, void* stream) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "invalid dtype");
  if (m < 1 || n < 1 || lda < m || m > 0x7fffffff || n > 0x7fffffff || !d_info)
    return fail(EIGAMD_ERR_ARG, "invalid LU arguments");
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  int fz = -1, nb = 0;
  HIP_TRY(getrf_device(dtype, m, n, d_a, lda, d_transp, max_block_size, &fz, &nb, static_cast<hipStream_t>(stream), true, d_info));
  return EIGAMD_OK;
}

int eigamd_laswp_dev(int dtype, void* d_a, int64_t lda, int64_t ncols, const int* d_transp, int k0, int k1,
                     void* stream) {
  if (dtype < 0 || dtype > 3 || lda < 1 || ncols < 0 || k0 < 0 || k1 < k0)
    return fail(EIGAMD_ERR_ARG, "invalid laswp arguments");
  if (ncols == 0 || k1 == k0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  HIP_TRY(laswp_device(dtype, d_a, lda, ncols, d_transp, k0, k1, static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

int eigamd_trsm_llnu_dev(int dtype, int64_t m, int64_t n, const void* d_l, int64_t ldl, void* d_b, int64_t ldb,
                         void* stream) {
  if (dtype < 0 || dtype > 3 || m < 0 || n < 0 || ldl < std::max<int64_t>(1, m) || ldb < std::max<int64_t>(1, m))
    return fail(EIGAMD_ERR_ARG, "invalid trsm arguments");
  if (m == 0 || n == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  HIP_TRY(trsm_llnu_device(dtype, m, n, d_l, ldl, d_b, ldb, static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

// Host-pointer LU: stage in, factorise on the device, copy factors and
// transpositions back.  Returns the index of the first zero pivot or -1 through
// *first_zero_pivot, like partial_lu_impl::blocked_lu (PartialPivLU.h:429).
// `inc`: distance between consecutive rows of a column (1 for a column-major matrix; a row-major matrix
// with row stride s is the case lda = 1, inc = s: the copy threads gather it into the dense column-major
// device image and scatter the factors back).
static int lu_host_impl(int dtype, int64_t m, int64_t n, void* a, int64_t lda, int* transp, int* first_zero_pivot,
                        int* nb_transpositions, int64_t inc = 1) {
  *first_zero_pivot = -1;
  *nb_transpositions = 0;
  if (m == 0 || n == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const int es = kElemSize[dtype];
  const int64_t dld = round_up(m, 16);
  const int64_t size = std::min(m, n);
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  if ((rc = st.init()) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(2, (size_t)dld * n * es)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(0, (size_t)size * sizeof(int) + 64)) != EIGAMD_OK) return rc;
  // The caller's matrix is still intact while the device works on its copy: when the one-launch
  // base case cannot run (its workgroups were not all resident, the device being shared with
  // other work), the copy is made afresh and factorised through the per-column kernels.
  for (int attempt = 0; attempt < 2; ++attempt) {
    if ((rc = st.upload(st.buf[2], dld, a, lda, m, n, es, inc)) != EIGAMD_OK) return rc;
    HIP_TRY(hipEventRecord(st.uploaded, st.h2d));
    HIP_TRY(hipStreamWaitEvent(st.stream, st.uploaded, 0));
    const hipError_t e = getrf_device(dtype, m, n, st.buf[2], dld, static_cast<int*>(st.buf[0]), 256, first_zero_pivot,
                                      nb_transpositions, st.stream, /*allow_fused=*/attempt == 0);
    if (e == kLuNotResident && attempt == 0) continue;
    if (e != hipSuccess) return hip_fail(e, "getrf_device");
    break;
  }
  HIP_TRY(hipMemcpyAsync(transp, st.buf[0], (size_t)size * sizeof(int), hipMemcpyDeviceToHost, st.stream));
  HIP_TRY(hipStreamSynchronize(st.stream));   // getrf_device has synchronised already; this covers the copy
  if ((rc = st.download(a, lda, st.buf[2], dld, m, n, es, inc)) != EIGAMD_OK) return rc;
  return EIGAMD_OK;
}

int eigamd_blocked_lu_host(int dtype, int64_t rows, int64_t cols, void* lu, int64_t lu_stride, int* row_transpositions,
                           int* nb_transpositions, int* first_zero_pivot) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "invalid dtype");
  if (rows < 0 || cols < 0 || lu_stride < std::max<int64_t>(1, rows) || rows > 0x7fffffff || cols > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid LU arguments");
  return lu_host_impl(dtype, rows, cols, lu, lu_stride, row_transpositions, first_zero_pivot, nb_transpositions);
}

int eigamd_blocked_lu_host_rowmajor(int dtype, int64_t rows, int64_t cols, void* lu, int64_t lu_stride, int* row_transpositions,
                                    int* nb_transpositions, int* first_zero_pivot) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "invalid dtype");
  if (rows < 0 || cols < 0 || lu_stride < std::max<int64_t>(1, cols) || rows > 0x7fffffff || cols > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid LU arguments");
  return lu_host_impl(dtype, rows, cols, lu, /*lda=*/1, row_transpositions, first_zero_pivot, nb_transpositions, /*inc=*/lu_stride);
}

static int f77_getrf(int dtype, const char* name, const int* m, const int* n, void* a, const int* lda, int* ipiv,
                     int* info) {
  *info = 0;                                   // lapack/lu.cpp:16-24
  if (*m < 0) *info = -1;
  else if (*n < 0) *info = -2;
  else if (*lda < std::max(1, *m)) *info = -4;
  if (*info != 0) {
    int e = -*info;
    return xerbla_(name, &e, 6);
  }
  if (*m == 0 || *n == 0) return 0;
  int fz = -1, nb = 0;
  int rc = lu_host_impl(dtype, *m, *n, a, *lda, ipiv, &fz, &nb);
  if (rc != EIGAMD_OK) {
    std::fprintf(stderr, "eigen_amd %s: %s\n", name, g_err.c_str());
    std::abort();
  }
  for (int i = 0; i < std::min(*m, *n); ++i) ipiv[i]++;   // 1-based (lu.cpp:34-35)
  if (fz >= 0) *info = fz + 1;                            // (lu.cpp:37-38)
  return 0;
}

int dgetrf_(const int* m, const int* n, double* a, const int* lda, int* ipiv, int* info) {
  return f77_getrf(F64, "DGETRF", m, n, a, lda, ipiv, info);
}
int sgetrf_(const int* m, const int* n, float* a, const int* lda, int* ipiv, int* info) {
  return f77_getrf(F32, "SGETRF", m, n, a, lda, ipiv, info);
}
int cgetrf_(const int* m, const int* n, void* a, const int* lda, int* ipiv, int* info) {   // lapack/complex_single.cpp
  return f77_getrf(C32, "CGETRF", m, n, a, lda, ipiv, info);
}
int zgetrf_(const int* m, const int* n, void* a, const int* lda, int* ipiv, int* info) {   // lapack/complex_double.cpp
  return f77_getrf(C64, "ZGETRF", m, n, a, lda, ipiv, info);
}

// ---- ?getrs_ (lapack/lu.cpp:46-89) ---------------------------------------------------

int eigamd_getrs_dev(int dtype, int op, int64_t n, int64_t nrhs, const void* d_lu, int64_t ldlu, const int* d_transp,
                     void* d_b, int64_t ldb, void* stream) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  if (op < OP_N || op > OP_C || n < 0 || nrhs < 0 || ldlu < std::max<int64_t>(1, n) || ldb < std::max<int64_t>(1, n) ||
      n > 0x7fffffff || nrhs > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid getrs arguments");
  if (n == 0 || nrhs == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  HIP_TRY(getrs_device(dtype, op, n, nrhs, d_lu, ldlu, d_transp, d_b, ldb, static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

// Host-pointer solve: factors, transpositions (0-based here) and right-hand sides staged in, the
// m x nrhs window of B copied back.
static int getrs_host_impl(int dtype, int op, int64_t n, int64_t nrhs, const void* lu, int64_t ldlu, const int* transp0,
                           void* b, int64_t ldb) {
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const int es = kElemSize[dtype];
  const int64_t dlda = round_up(n, 16);
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  if ((rc = st.init()) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(0, (size_t)dlda * n * es)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(1, (size_t)n * sizeof(int) + 64)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(2, (size_t)dlda * nrhs * es)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[0], dlda, lu, ldlu, n, n, es)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[2], dlda, b, ldb, n, nrhs, es)) != EIGAMD_OK) return rc;
  HIP_TRY(hipEventRecord(st.uploaded, st.h2d));
  HIP_TRY(hipStreamWaitEvent(st.stream, st.uploaded, 0));
  HIP_TRY(hipMemcpyAsync(st.buf[1], transp0, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st.stream));
  HIP_TRY(getrs_device(dtype, op, n, nrhs, st.buf[0], dlda, static_cast<const int*>(st.buf[1]), st.buf[2], dlda, st.stream));
  HIP_TRY(hipEventRecord(st.computed, st.stream));
  HIP_TRY(hipStreamWaitEvent(st.d2h, st.computed, 0));
  return st.download(b, ldb, st.buf[2], dlda, n, nrhs, es);
}

static int f77_getrs(int dtype, const char* name, const char* trans, const int* n, const int* nrhs, const void* a,
                     const int* lda, int* ipiv, void* b, const int* ldb, int* info) {
  const int op = op_of(*trans);
  *info = 0;                                   // lapack/lu.cpp:48-58
  if (op < 0) *info = -1;
  else if (*n < 0) *info = -2;
  else if (*nrhs < 0) *info = -3;
  else if (*lda < std::max(1, *n)) *info = -5;
  else if (*ldb < std::max(1, *n)) *info = -8;
  if (*info != 0) {
    int e = -*info;
    return xerbla_(name, &e, 6);
  }
  if (*n == 0 || *nrhs == 0) return 0;         // every statement of lu.cpp:65-86 is empty then
  std::vector<int> transp0(ipiv, ipiv + *n);   // the reference shifts ipiv in place and back (:65-66, :85-86)
  for (int& t : transp0) --t;
  for (int i = 0; i < *n; ++i)
    if (transp0[i] < 0 || transp0[i] >= *n) {
      std::fprintf(stderr, "eigen_amd %s: ipiv[%d] = %d is not a row of the matrix\n", name, i, ipiv[i]);
      std::abort();
    }
  int rc = getrs_host_impl(dtype, op, *n, *nrhs, a, *lda, transp0.data(), b, *ldb);
  if (rc != EIGAMD_OK) {
    std::fprintf(stderr, "eigen_amd %s: %s\n", name, g_err.c_str());
    std::abort();
  }
  return 0;
}

int sgetrs_(const char* trans, const int* n, const int* nrhs, const float* a, const int* lda, int* ipiv, float* b,
            const int* ldb, int* info) {
  return f77_getrs(F32, "SGETRS", trans, n, nrhs, a, lda, ipiv, b, ldb, info);
}
int dgetrs_(const char* trans, const int* n, const int* nrhs, const double* a, const int* lda, int* ipiv, double* b,
            const int* ldb, int* info) {
  return f77_getrs(F64, "DGETRS", trans, n, nrhs, a, lda, ipiv, b, ldb, info);
}
int cgetrs_(const char* trans, const int* n, const int* nrhs, const void* a, const int* lda, int* ipiv, void* b,
            const int* ldb, int* info) {
  return f77_getrs(C32, "CGETRS", trans, n, nrhs, a, lda, ipiv, b, ldb, info);
}
int zgetrs_(const char* trans, const int* n, const int* nrhs, const void* a, const int* lda, int* ipiv, void* b,
            const int* ldb, int* info) {
  return f77_getrs(C64, "ZGETRS", trans, n, nrhs, a, lda, ipiv, b, ldb, info);
}

// ---- ?trsm_ (blas/level3_impl.h:78-178) ---------------------------------------------

static int side_of(char c) { return (c == 'L' || c == 'l') ? 0 : (c == 'R' || c == 'r') ? 1 : -1; }   // common.h SIDE
static int uplo_of(char c) { return (c == 'U' || c == 'u') ? 0 : (c == 'L' || c == 'l') ? 1 : -1; }   // common.h UPLO
static int diag_of(char c) { return (c == 'N' || c == 'n') ? 0 : (c == 'U' || c == 'u') ? 1 : -1; }   // common.h DIAG

int eigamd_trsm_dev(int dtype, int side, int uplo, int op, int diag, int64_t m, int64_t n, const void* alpha,
                    const void* d_a, int64_t lda, void* d_b, int64_t ldb, void* stream) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  if (side < 0 || side > 1 || uplo < 0 || uplo > 1 || op < OP_N || op > OP_C || diag < 0 || diag > 1 || m < 0 || n < 0 ||
      lda < std::max<int64_t>(1, side == 0 ? m : n) || ldb < std::max<int64_t>(1, m) || m > 0x7fffffff || n > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid trsm arguments");
  if (m == 0 || n == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  double al[2];
  read_scalar(dtype, alpha, al);
  HIP_TRY(trsm_device(dtype, side == 0, uplo == 0, op, diag == 1, m, n, al, d_a, lda, d_b, ldb,
                      static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

static int trsm_host_impl(int dtype, int side, int uplo, int op, int diag, int64_t m, int64_t n, const double alpha[2],
                          const void* a, int64_t lda, void* b, int64_t ldb) {
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const int es = kElemSize[dtype];
  const int64_t na = side == 0 ? m : n;
  const int64_t dlda = round_up(na, 16), dldb = round_up(m, 16);
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  if ((rc = st.init()) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(0, (size_t)dlda * na * es)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(2, (size_t)dldb * n * es)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[0], dlda, a, lda, na, na, es)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[2], dldb, b, ldb, m, n, es)) != EIGAMD_OK) return rc;
  HIP_TRY(hipEventRecord(st.uploaded, st.h2d));
  HIP_TRY(hipStreamWaitEvent(st.stream, st.uploaded, 0));
  HIP_TRY(trsm_device(dtype, side == 0, uplo == 0, op, diag == 1, m, n, alpha, st.buf[0], dlda, st.buf[2], dldb, st.stream));
  // only the m x n window goes back
  HIP_TRY(hipEventRecord(st.computed, st.stream));
  HIP_TRY(hipStreamWaitEvent(st.d2h, st.computed, 0));
  return st.download(b, ldb, st.buf[2], dldb, m, n, es);
}

int eigamd_trsm_host(int dtype, int side, int uplo, int op, int diag, int64_t m, int64_t n, const void* alpha,
                     const void* a, int64_t lda, void* b, int64_t ldb) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  if (side < 0 || side > 1 || uplo < 0 || uplo > 1 || op < OP_N || op > OP_C || diag < 0 || diag > 1 || m < 0 || n < 0 ||
      lda < std::max<int64_t>(1, side == 0 ? m : n) || ldb < std::max<int64_t>(1, m) || m > 0x7fffffff || n > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid trsm arguments");
  if (m == 0 || n == 0) return EIGAMD_OK;
  double al[2];
  read_scalar(dtype, alpha, al);
  return trsm_host_impl(dtype, side, uplo, op, diag, m, n, al, a, lda, b, ldb);
}

static int f77_trsm(int dtype, const char* name, const char* side, const char* uplo, const char* opa, const char* diag,
                    const int* m, const int* n, const void* alpha, const void* a, const int* lda, void* b,
                    const int* ldb) {
  const int sd = side_of(*side), ul = uplo_of(*uplo), op = op_of(*opa), dg = diag_of(*diag);
  int info = 0;                                                    // level3_impl.h:146-156
  if (sd < 0) info = 1;
  else if (ul < 0) info = 2;
  else if (op < 0) info = 3;
  else if (dg < 0) info = 4;
  else if (*m < 0) info = 5;
  else if (*n < 0) info = 6;
  else if (*lda < std::max(1, sd == 0 ? *m : *n)) info = 9;
  else if (*ldb < std::max(1, *m)) info = 11;
  if (info) return xerbla_(name, &info, 6);
  if (*m == 0 || *n == 0) return 0;                                // level3_impl.h:158-159
  double al[2];
  read_scalar(dtype, alpha, al);
  int rc = trsm_host_impl(dtype, sd, ul, op, dg, *m, *n, al, a, *lda, b, *ldb);
  if (rc != EIGAMD_OK) {
    std::fprintf(stderr, "eigen_amd %s: %s\n", name, g_err.c_str());
    std::abort();
  }
  return 0;
}

int strsm_(const char* side, const char* uplo, const char* opa, const char* diag, const int* m, const int* n,
           const float* alpha, const float* a, const int* lda, float* b, const int* ldb) {
  return f77_trsm(F32, "STRSM ", side, uplo, opa, diag, m, n, alpha, a, lda, b, ldb);
}
int dtrsm_(const char* side, const char* uplo, const char* opa, const char* diag, const int* m, const int* n,
           const double* alpha, const double* a, const int* lda, double* b, const int* ldb) {
  return f77_trsm(F64, "DTRSM ", side, uplo, opa, diag, m, n, alpha, a, lda, b, ldb);
}
int ctrsm_(const char* side, const char* uplo, const char* opa, const char* diag, const int* m, const int* n,
           const float* alpha, const float* a, const int* lda, float* b, const int* ldb) {
  return f77_trsm(C32, "CTRSM ", side, uplo, opa, diag, m, n, alpha, a, lda, b, ldb);
}
int ztrsm_(const char* side, const char* uplo, const char* opa, const char* diag, const int* m, const int* n,
           const double* alpha, const double* a, const int* lda, double* b, const int* ldb) {
  return f77_trsm(C64, "ZTRSM ", side, uplo, opa, diag, m, n, alpha, a, lda, b, ldb);
}

// ---- ?syrk_ (blas/level3_impl.h:357-433) ---------------------------------------------

int eigamd_syrk_dev(int dtype, int uplo, int op, int64_t n, int64_t k, const void* alpha, const void* d_a, int64_t lda,
                    const void* beta, void* d_c, int64_t ldc, void* stream) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  if (uplo < 0 || uplo > 1 || (op != OP_N && op != OP_T) || n < 0 || k < 0 ||
      lda < std::max<int64_t>(1, op == OP_N ? n : k) || ldc < std::max<int64_t>(1, n) || n > 0x7fffffff || k > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid syrk arguments");
  if (n == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  double al[2], be[2];
  read_scalar(dtype, alpha, al);
  read_scalar(dtype, beta, be);
  HIP_TRY(syrk_device(dtype, uplo == 0, op, n, k, al, d_a, lda, be, d_c, ldc, static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

static int syrk_host_impl(int dtype, int uplo, int op, int64_t n, int64_t k, const double alpha[2], const void* a,
                          int64_t lda, const double beta[2], void* c, int64_t ldc) {
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const int es = kElemSize[dtype];
  const int64_t a_rows = op == OP_N ? n : k, a_cols = op == OP_N ? k : n;
  const int64_t dlda = round_up(std::max<int64_t>(a_rows, 1), 16), dldc = round_up(n, 16);
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  if ((rc = st.init()) != EIGAMD_OK) return rc;
  if (k > 0 && (rc = st.reserve(0, (size_t)dlda * a_cols * es)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(2, (size_t)dldc * n * es)) != EIGAMD_OK) return rc;
  if (k > 0 && (rc = st.upload(st.buf[0], dlda, a, lda, a_rows, a_cols, es)) != EIGAMD_OK) return rc;
  // the whole n x n window travels both ways; the kernels touch the uplo triangle only, so
  // the other triangle comes back with the bytes it went out with
  if ((rc = st.upload(st.buf[2], dldc, c, ldc, n, n, es)) != EIGAMD_OK) return rc;
  HIP_TRY(hipEventRecord(st.uploaded, st.h2d));
  HIP_TRY(hipStreamWaitEvent(st.stream, st.uploaded, 0));
  HIP_TRY(syrk_device(dtype, uplo == 0, op, n, k, alpha, st.buf[0], dlda, beta, st.buf[2], dldc, st.stream));
  HIP_TRY(hipEventRecord(st.computed, st.stream));
  HIP_TRY(hipStreamWaitEvent(st.d2h, st.computed, 0));
  return st.download(c, ldc, st.buf[2], dldc, n, n, es);
}

static int f77_syrk(int dtype, const char* name, const char* uplo, const char* opc, const int* n, const int* k,
                    const void* alpha, const void* a, const int* lda, const void* beta, void* c, const int* ldc) {
  const int ul = uplo_of(*uplo);
  int op = op_of(*opc);
  int info = 0;                                                    // level3_impl.h:386-394
  if (ul < 0) info = 1;
  else if (op < 0 || (dtype >= C32 && op == OP_C)) info = 2;
  else if (*n < 0) info = 3;
  else if (*k < 0) info = 4;
  else if (*lda < std::max(1, op == OP_N ? *n : *k)) info = 7;
  else if (*ldc < std::max(1, *n)) info = 10;
  if (info) return xerbla_(name, &info, 6);
  if (op == OP_C) op = OP_T;   // real kinds: 'C' is 'T'
  if (*n == 0) return 0;
  double al[2], be[2];
  read_scalar(dtype, alpha, al);
  read_scalar(dtype, beta, be);
  if (*k == 0 && be[0] == 1.0 && be[1] == 0.0) return 0;          // level3_impl.h:396,406
  int rc = syrk_host_impl(dtype, ul, op, *n, *k, al, a, *lda, be, c, *ldc);
  if (rc != EIGAMD_OK) {
    std::fprintf(stderr, "eigen_amd %s: %s\n", name, g_err.c_str());
    std::abort();
  }
  return 0;
}

int ssyrk_(const char* uplo, const char* trans, const int* n, const int* k, const float* alpha, const float* a,
           const int* lda, const float* beta, float* c, const int* ldc) {
  return f77_syrk(F32, "SSYRK ", uplo, trans, n, k, alpha, a, lda, beta, c, ldc);
}
int dsyrk_(const char* uplo, const char* trans, const int* n, const int* k, const double* alpha, const double* a,
           const int* lda, const double* beta, double* c, const int* ldc) {
  return f77_syrk(F64, "DSYRK ", uplo, trans, n, k, alpha, a, lda, beta, c, ldc);
}
int csyrk_(const char* uplo, const char* trans, const int* n, const int* k, const float* alpha, const float* a,
           const int* lda, const float* beta, float* c, const int* ldc) {
  return f77_syrk(C32, "CSYRK ", uplo, trans, n, k, alpha, a, lda, beta, c, ldc);
}
int zsyrk_(const char* uplo, const char* trans, const int* n, const int* k, const double* alpha, const double* a,
           const int* lda, const double* beta, double* c, const int* ldc) {
  return f77_syrk(C64, "ZSYRK ", uplo, trans, n, k, alpha, a, lda, beta, c, ldc);
}

// ---- ?herk_ (blas/level3_impl.h:562-624) ------------------------------------------------

static double read_real(int dtype, const void* p) {
  return (dtype == F32 || dtype == C32) ? (double)*static_cast<const float*>(p) : *static_cast<const double*>(p);
}

int eigamd_herk_dev(int dtype, int uplo, int op, int64_t n, int64_t k, const void* alpha, const void* d_a, int64_t lda,
                    const void* beta, void* d_c, int64_t ldc, void* stream) {
  if (dtype != C32 && dtype != C64) return fail(EIGAMD_ERR_ARG, "herk is defined for the complex kinds");
  if (uplo < 0 || uplo > 1 || (op != OP_N && op != OP_C) || n < 0 || k < 0 ||
      lda < std::max<int64_t>(1, op == OP_N ? n : k) || ldc < std::max<int64_t>(1, n) || n > 0x7fffffff || k > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid herk arguments");
  if (n == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  HIP_TRY(herk_device(dtype, uplo == 0, op, n, k, read_real(dtype, alpha), d_a, lda, read_real(dtype, beta), d_c, ldc,
                      static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

static int herk_host_impl(int dtype, int uplo, int op, int64_t n, int64_t k, double alpha, const void* a, int64_t lda,
                          double beta, void* c, int64_t ldc) {
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const int es = kElemSize[dtype];
  const int64_t a_rows = op == OP_N ? n : k, a_cols = op == OP_N ? k : n;
  const int64_t dlda = round_up(std::max<int64_t>(a_rows, 1), 16), dldc = round_up(n, 16);
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  if ((rc = st.init()) != EIGAMD_OK) return rc;
  if (k > 0 && (rc = st.reserve(0, (size_t)dlda * a_cols * es)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(2, (size_t)dldc * n * es)) != EIGAMD_OK) return rc;
  if (k > 0 && (rc = st.upload(st.buf[0], dlda, a, lda, a_rows, a_cols, es)) != EIGAMD_OK) return rc;
  // the whole n x n window travels both ways; the kernels touch the uplo triangle only
  if ((rc = st.upload(st.buf[2], dldc, c, ldc, n, n, es)) != EIGAMD_OK) return rc;
  HIP_TRY(hipEventRecord(st.uploaded, st.h2d));
  HIP_TRY(hipStreamWaitEvent(st.stream, st.uploaded, 0));
  HIP_TRY(herk_device(dtype, uplo == 0, op, n, k, alpha, st.buf[0], dlda, beta, st.buf[2], dldc, st.stream));
  HIP_TRY(hipEventRecord(st.computed, st.stream));
  HIP_TRY(hipStreamWaitEvent(st.d2h, st.computed, 0));
  return st.download(c, ldc, st.buf[2], dldc, n, n, es);
}

static int f77_herk(int dtype, const char* name, const char* uplo, const char* opc, const int* n, const int* k,
                    const void* alpha, const void* a, const int* lda, const void* beta, void* c, const int* ldc) {
  const int ul = uplo_of(*uplo), op = op_of(*opc);
  int info = 0;                                                    // level3_impl.h:591-599
  if (ul < 0) info = 1;
  else if (op < 0 || op == OP_T) info = 2;
  else if (*n < 0) info = 3;
  else if (*k < 0) info = 4;
  else if (*lda < std::max(1, op == OP_N ? *n : *k)) info = 7;
  else if (*ldc < std::max(1, *n)) info = 10;
  if (info) return xerbla_(name, &info, 6);
  if (*n == 0) return 0;
  const double al = read_real(dtype, alpha), be = read_real(dtype, beta);
  if ((*k == 0 || al == 0.0) && be == 1.0) return 0;               // nothing to do: C keeps every bit
  int rc = herk_host_impl(dtype, ul, op, *n, *k, al, a, *lda, be, c, *ldc);
  if (rc != EIGAMD_OK) {
    std::fprintf(stderr, "eigen_amd %s: %s\n", name, g_err.c_str());
    std::abort();
  }
  return 0;
}

int cherk_(const char* uplo, const char* trans, const int* n, const int* k, const float* alpha, const float* a,
           const int* lda, const float* beta, float* c, const int* ldc) {
  return f77_herk(C32, "CHERK ", uplo, trans, n, k, alpha, a, lda, beta, c, ldc);
}
int zherk_(const char* uplo, const char* trans, const int* n, const int* k, const double* alpha, const double* a,
           const int* lda, const double* beta, double* c, const int* ldc) {
  return f77_herk(C64, "ZHERK ", uplo, trans, n, k, alpha, a, lda, beta, c, ldc);
}

// ---- ?syr2k_ / ?her2k_ (blas/level3_impl.h:435-501, :626-700) -----------------------------

int eigamd_syr2k_dev(int dtype, int uplo, int op, int hermitian, int64_t n, int64_t k, const void* alpha, const void* d_a,
                     int64_t lda, const void* d_b, int64_t ldb, const void* beta, void* d_c, int64_t ldc, void* stream) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  if (hermitian && dtype < C32) return fail(EIGAMD_ERR_ARG, "her2k is defined for the complex kinds");
  const bool op_ok = op == OP_N || (hermitian ? op == OP_C : op == OP_T);
  const int64_t need = std::max<int64_t>(1, op == OP_N ? n : k);
  if (uplo < 0 || uplo > 1 || !op_ok || n < 0 || k < 0 || lda < need || ldb < need || ldc < std::max<int64_t>(1, n) ||
      n > 0x7fffffff || k > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid syr2k arguments");
  if (n == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  double al[2], be[2];
  read_scalar(dtype, alpha, al);
  read_scalar(dtype, beta, be);
  if (hermitian && be[1] != 0.0) return fail(EIGAMD_ERR_ARG, "her2k takes a real beta");
  HIP_TRY(syr2k_device(dtype, uplo == 0, op, hermitian != 0, n, k, al, d_a, lda, d_b, ldb, be, d_c, ldc,
                       static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

static int syr2k_host_impl(int dtype, int uplo, int op, bool herm, int64_t n, int64_t k, const double alpha[2], const void* a,
                           int64_t lda, const void* b, int64_t ldb, const double beta[2], void* c, int64_t ldc) {
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const int es = kElemSize[dtype];
  const int64_t rows = op == OP_N ? n : k, cols = op == OP_N ? k : n;
  const int64_t dld = round_up(std::max<int64_t>(rows, 1), 16), dldc = round_up(n, 16);
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  if ((rc = st.init()) != EIGAMD_OK) return rc;
  if (k > 0 && (rc = st.reserve(0, (size_t)dld * cols * es)) != EIGAMD_OK) return rc;
  if (k > 0 && (rc = st.reserve(1, (size_t)dld * cols * es)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(2, (size_t)dldc * n * es)) != EIGAMD_OK) return rc;
  if (k > 0 && (rc = st.upload(st.buf[0], dld, a, lda, rows, cols, es)) != EIGAMD_OK) return rc;
  if (k > 0 && (rc = st.upload(st.buf[1], dld, b, ldb, rows, cols, es)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[2], dldc, c, ldc, n, n, es)) != EIGAMD_OK) return rc;
  HIP_TRY(hipEventRecord(st.uploaded, st.h2d));
  HIP_TRY(hipStreamWaitEvent(st.stream, st.uploaded, 0));
  HIP_TRY(syr2k_device(dtype, uplo == 0, op, herm, n, k, alpha, st.buf[0], dld, st.buf[1], dld, beta, st.buf[2], dldc, st.stream));
  HIP_TRY(hipEventRecord(st.computed, st.stream));
  HIP_TRY(hipStreamWaitEvent(st.d2h, st.computed, 0));
  return st.download(c, ldc, st.buf[2], dldc, n, n, es);
}

static int f77_syr2k(int dtype, bool herm, const char* name, const char* uplo, const char* opc, const int* n, const int* k,
                     const void* alpha, const void* a, const int* lda, const void* b, const int* ldb, const void* beta,
                     void* c, const int* ldc) {
  const int ul = uplo_of(*uplo);
  int op = op_of(*opc);
  int info = 0;                                                    // level3_impl.h:450-459 / :642-651
  if (ul < 0) info = 1;
  else if (op < 0 || (herm ? op == OP_T : (dtype >= C32 && op == OP_C))) info = 2;
  else if (*n < 0) info = 3;
  else if (*k < 0) info = 4;
  else if (*lda < std::max(1, op == OP_N ? *n : *k)) info = 7;
  else if (*ldb < std::max(1, op == OP_N ? *n : *k)) info = 9;
  else if (*ldc < std::max(1, *n)) info = 12;
  if (info) return xerbla_(name, &info, 6);
  if (!herm && op == OP_C) op = OP_T;   // real kinds: 'C' is 'T'
  const int ret = herm ? 1 : (*k == 0 ? 1 : 0);                    // what the reference returns (:474-475, :699)
  if (*n == 0) return ret;
  double al[2], be[2];
  read_scalar(dtype, alpha, al);
  if (herm) { be[0] = read_real(dtype, beta); be[1] = 0.0; }
  else read_scalar(dtype, beta, be);
  if ((*k == 0 || (al[0] == 0.0 && al[1] == 0.0)) && be[0] == 1.0 && be[1] == 0.0) return ret;
  int rc = syr2k_host_impl(dtype, ul, op, herm, *n, *k, al, a, *lda, b, *ldb, be, c, *ldc);
  if (rc != EIGAMD_OK) {
    std::fprintf(stderr, "eigen_amd %s: %s\n", name, g_err.c_str());
    std::abort();
  }
  return ret;
}

#define EIGAMD_SYR2K_ENTRY(fn, T, kind, herm, name)                                                                       \
  int fn(const char* uplo, const char* trans, const int* n, const int* k, const T* alpha, const T* a, const int* lda,      \
         const T* b, const int* ldb, const T* beta, T* c, const int* ldc) {                                                \
    return f77_syr2k(kind, herm, name, uplo, trans, n, k, alpha, a, lda, b, ldb, beta, c, ldc);                            \
  }
EIGAMD_SYR2K_ENTRY(ssyr2k_, float, F32, false, "SSYR2K")
EIGAMD_SYR2K_ENTRY(dsyr2k_, double, F64, false, "DSYR2K")
EIGAMD_SYR2K_ENTRY(csyr2k_, float, C32, false, "CSYR2K")
EIGAMD_SYR2K_ENTRY(zsyr2k_, double, C64, false, "ZSYR2K")
EIGAMD_SYR2K_ENTRY(cher2k_, float, C32, true, "CHER2K")
EIGAMD_SYR2K_ENTRY(zher2k_, double, C64, true, "ZHER2K")

// ---- ?symm_ / ?hemm_ (blas/level3_impl.h:245-353, :505-560) and ?trmm_ (:183-243) -------

int eigamd_symm_dev(int dtype, int side, int uplo, int hermitian, int64_t m, int64_t n, const void* alpha, const void* d_a,
                    int64_t lda, const void* d_b, int64_t ldb, const void* beta, void* d_c, int64_t ldc, void* stream) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  if (side < 0 || side > 1 || uplo < 0 || uplo > 1 || m < 0 || n < 0 || lda < std::max<int64_t>(1, side == 0 ? m : n) ||
      ldb < std::max<int64_t>(1, m) || ldc < std::max<int64_t>(1, m) || m > 0x7fffffff || n > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid symm arguments");
  if (m == 0 || n == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  double al[2], be[2];
  read_scalar(dtype, alpha, al);
  read_scalar(dtype, beta, be);
  HIP_TRY(symm_device(dtype, side == 0, uplo == 0, hermitian != 0 && dtype >= C32, m, n, al, d_a, lda, d_b, ldb, be, d_c, ldc,
                      static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

int eigamd_trmm_dev(int dtype, int side, int uplo, int op, int diag, int64_t m, int64_t n, const void* alpha,
                    const void* d_a, int64_t lda, void* d_b, int64_t ldb, void* stream) {
  if (dtype < 0 || dtype > 3) return fail(EIGAMD_ERR_ARG, "dtype out of range");
  if (side < 0 || side > 1 || uplo < 0 || uplo > 1 || op < OP_N || op > OP_C || diag < 0 || diag > 1 || m < 0 || n < 0 ||
      lda < std::max<int64_t>(1, side == 0 ? m : n) || ldb < std::max<int64_t>(1, m) || m > 0x7fffffff || n > 0x7fffffff)
    return fail(EIGAMD_ERR_ARG, "invalid trmm arguments");
  if (m == 0 || n == 0) return EIGAMD_OK;
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  double al[2];
  read_scalar(dtype, alpha, al);
  HIP_TRY(trmm_device(dtype, side == 0, uplo == 0, op, diag == 1, m, n, al, d_a, lda, d_b, ldb,
                      static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

// Host-pointer forms: stage A (na x na), B and C / B; only the m x n window comes back.
static int symm_host_impl(int dtype, int side, int uplo, bool herm, int64_t m, int64_t n, const double alpha[2],
                          const void* a, int64_t lda, const void* b, int64_t ldb, const double beta[2], void* c,
                          int64_t ldc) {
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const int es = kElemSize[dtype];
  const int64_t na = side == 0 ? m : n;
  const int64_t dlda = round_up(na, 16), dldb = round_up(m, 16), dldc = round_up(m, 16);
  const bool beta_is_zero = beta[0] == 0.0 && beta[1] == 0.0;
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  if ((rc = st.init()) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(0, (size_t)dlda * na * es)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(1, (size_t)dldb * n * es)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(2, (size_t)dldc * n * es)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[0], dlda, a, lda, na, na, es)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[1], dldb, b, ldb, m, n, es)) != EIGAMD_OK) return rc;
  if (!beta_is_zero && (rc = st.upload(st.buf[2], dldc, c, ldc, m, n, es)) != EIGAMD_OK) return rc;
  HIP_TRY(hipEventRecord(st.uploaded, st.h2d));
  HIP_TRY(hipStreamWaitEvent(st.stream, st.uploaded, 0));
  HIP_TRY(symm_device(dtype, side == 0, uplo == 0, herm, m, n, alpha, st.buf[0], dlda, st.buf[1], dldb, beta, st.buf[2], dldc,
                      st.stream));
  HIP_TRY(hipEventRecord(st.computed, st.stream));
  HIP_TRY(hipStreamWaitEvent(st.d2h, st.computed, 0));
  return st.download(c, ldc, st.buf[2], dldc, m, n, es);
}

static int f77_symm(int dtype, bool herm, const char* name, const char* side, const char* uplo, const int* m, const int* n,
                    const void* alpha, const void* a, const int* lda, const void* b, const int* ldb, const void* beta,
                    void* c, const int* ldc) {
  const int sd = side_of(*side), ul = uplo_of(*uplo);
  int info = 0;                                                    // level3_impl.h:256-266 / :517-527
  if (sd < 0) info = 1;
  else if (ul < 0) info = 2;
  else if (*m < 0) info = 3;
  else if (*n < 0) info = 4;
  else if (*lda < std::max(1, sd == 0 ? *m : *n)) info = 7;
  else if (*ldb < std::max(1, *m)) info = 9;
  else if (*ldc < std::max(1, *m)) info = 12;
  if (info) return xerbla_(name, &info, 6);
  if (*m == 0 || *n == 0) return 0;
  double al[2], be[2];
  read_scalar(dtype, alpha, al);
  read_scalar(dtype, beta, be);
  int rc = symm_host_impl(dtype, sd, ul, herm, *m, *n, al, a, *lda, b, *ldb, be, c, *ldc);
  if (rc != EIGAMD_OK) {
    std::fprintf(stderr, "eigen_amd %s: %s\n", name, g_err.c_str());
    std::abort();
  }
  return 0;
}

#define EIGAMD_SYMM_ENTRY(fn, T, kind, herm, name)                                                                      \
  int fn(const char* side, const char* uplo, const int* m, const int* n, const T* alpha, const T* a, const int* lda,      \
         const T* b, const int* ldb, const T* beta, T* c, const int* ldc) {                                               \
    return f77_symm(kind, herm, name, side, uplo, m, n, alpha, a, lda, b, ldb, beta, c, ldc);                             \
  }
EIGAMD_SYMM_ENTRY(ssymm_, float, F32, false, "SSYMM ")
EIGAMD_SYMM_ENTRY(dsymm_, double, F64, false, "DSYMM ")
EIGAMD_SYMM_ENTRY(csymm_, float, C32, false, "CSYMM ")
EIGAMD_SYMM_ENTRY(zsymm_, double, C64, false, "ZSYMM ")
EIGAMD_SYMM_ENTRY(chemm_, float, C32, true, "CHEMM ")
EIGAMD_SYMM_ENTRY(zhemm_, double, C64, true, "ZHEMM ")

static int trmm_host_impl(int dtype, int side, int uplo, int op, int diag, int64_t m, int64_t n, const double alpha[2],
                          const void* a, int64_t lda, void* b, int64_t ldb) {
  int rc = ensure_device();
  if (rc != EIGAMD_OK) return rc;
  const int es = kElemSize[dtype];
  const int64_t na = side == 0 ? m : n;
  const int64_t dlda = round_up(na, 16), dldb = round_up(m, 16);
  Staging& st = staging();
  std::lock_guard<std::mutex> lock(st.mu);
  if ((rc = st.init()) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(0, (size_t)dlda * na * es)) != EIGAMD_OK) return rc;
  if ((rc = st.reserve(2, (size_t)dldb * n * es)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[0], dlda, a, lda, na, na, es)) != EIGAMD_OK) return rc;
  if ((rc = st.upload(st.buf[2], dldb, b, ldb, m, n, es)) != EIGAMD_OK) return rc;
  HIP_TRY(hipEventRecord(st.uploaded, st.h2d));
  HIP_TRY(hipStreamWaitEvent(st.stream, st.uploaded, 0));
  HIP_TRY(trmm_device(dtype, side == 0, uplo == 0, op, diag == 1, m, n, alpha, st.buf[0], dlda, st.buf[2], dldb, st.stream));
  HIP_TRY(hipEventRecord(st.computed, st.stream));
  HIP_TRY(hipStreamWaitEvent(st.d2h, st.computed, 0));
  return st.download(b, ldb, st.buf[2], dldb, m, n, es);
}

static int f77_trmm(int dtype, const char* name, const char* side, const char* uplo, const char* opa, const char* diag,
                    const int* m, const int* n, const void* alpha, const void* a, const int* lda, void* b,
                    const int* ldb) {
  const int sd = side_of(*side), ul = uplo_of(*uplo), op = op_of(*opa), dg = diag_of(*diag);
  int info = 0;                                                    // level3_impl.h:213-223
  if (sd < 0) info = 1;
  else if (ul < 0) info = 2;
  else if (op < 0) info = 3;
  else if (dg < 0) info = 4;
  else if (*m < 0) info = 5;
  else if (*n < 0) info = 6;
  else if (*lda < std::max(1, sd == 0 ? *m : *n)) info = 9;
  else if (*ldb < std::max(1, *m)) info = 11;
  if (info) return xerbla_(name, &info, 6);
  if (*m == 0 || *n == 0) return 1;                                // the reference returns 1 here (:227-228)
  double al[2];
  read_scalar(dtype, alpha, al);
  int rc = trmm_host_impl(dtype, sd, ul, op, dg, *m, *n, al, a, *lda, b, *ldb);
  if (rc != EIGAMD_OK) {
    std::fprintf(stderr, "eigen_amd %s: %s\n", name, g_err.c_str());
    std::abort();
  }
  return 1;
}

#define EIGAMD_TRMM_ENTRY(fn, T, kind, name)                                                                              \
  int fn(const char* side, const char* uplo, const char* opa, const char* diag, const int* m, const int* n, const T* alpha, \
         const T* a, const int* lda, T* b, const int* ldb) {                                                                \
    return f77_trmm(kind, name, side, uplo, opa, diag, m, n, alpha, a, lda, b, ldb);                                        \
  }
EIGAMD_TRMM_ENTRY(strmm_, float, F32, "STRMM ")
EIGAMD_TRMM_ENTRY(dtrmm_, double, F64, "DTRMM ")
EIGAMD_TRMM_ENTRY(ctrmm_, float, C32, "CTRMM ")
EIGAMD_TRMM_ENTRY(ztrmm_, double, C64, "ZTRMM ")

const char* eigamd_last_error(void) { return g_err.c_str(); }

int eigamd_device_ok(void) { return ensure_device() == EIGAMD_OK ? 1 : 0; }

int eigamd_gemm_plan(int dtype, int64_t m, int64_t n, int64_t k, int c_is_an_operand, int* plan) {
  if (dtype < 0 || dtype > 3 || m < 1 || n < 1 || k < 1 || !plan) return fail(EIGAMD_ERR_ARG, "invalid plan query");
  static const char dummy[16] = {0};
  GemmCall c;
  c.kind = dtype;
  c.opa = OP_N; c.opb = OP_N;
  c.m = m; c.n = n; c.k = k;
  c.alpha[0] = 1.0; c.alpha[1] = 0.0;
  c.beta[0] = 1.0; c.beta[1] = 0.0;
  c.A = dummy; c.B = dummy + 8; c.C = c_is_an_operand ? const_cast<char*>(dummy + 8) : const_cast<char*>(dummy + 4);  // (never dereferenced)
  c.lda = m; c.ldb = k; c.ldc = m;
  switch (dtype) {
    case F32: describe_gemm_f32(c, plan); break;
    case F64: describe_gemm_f64(c, plan); break;
    case C32: describe_gemm_c32(c, plan); break;
    default:  describe_gemm_c64(c, plan); break;
  }
  return EIGAMD_OK;
}

void eigamd_tile_shape(int dtype, int* bm, int* bn, int* bk) {
  switch (dtype) {
    case F32: *bm = Geometry<F32>::BM; *bn = Geometry<F32>::BN; *bk = Geometry<F32>::BK; break;
    case F64: *bm = Geometry<F64>::BM; *bn = Geometry<F64>::BN; *bk = Geometry<F64>::BK; break;
    case C32: *bm = Geometry<C32>::BM; *bn = Geometry<C32>::BN; *bk = Geometry<C32>::BK; break;
    default:  *bm = Geometry<C64>::BM; *bn = Geometry<C64>::BN; *bk = Geometry<C64>::BK; break;
  }
}

}  // extern "C"
