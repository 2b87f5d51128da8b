// Probe: asynchronous 2-D copies between ONE pageable host matrix and device memory, issued for neighbouring
// blocks on several streams at once -- the pattern the multi-device host route used before its copies went
// through pinned staging chunks.  Every HIP call's status is printed on failure (no abort of our own), the data
// is checked after each round, and the phases are:  1 = one host thread issues everything,  2 = one host
// thread per stream.  Usage: pageable_copy_probe [n=8192] [block=2048] [streams=4] [rounds=10]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(expr)                                                                                         \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      std::fprintf(stderr, "FAILED %s -> %d (%s) at line %d\n", #expr, (int)e_, hipGetErrorString(e_), __LINE__); \
      std::fflush(stderr);                                                                               \
      failed = true;                                                                                     \
    }                                                                                                    \
  } while (0)

static bool failed = false;

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 8192, nb = argc > 2 ? atol(argv[2]) : 2048;
  const int ns = argc > 3 ? atoi(argv[3]) : 4, rounds = argc > 4 ? atoi(argv[4]) : 10;
  const long ld = n + 8, blocks = (n + nb - 1) / nb;
  std::vector<double> host((size_t)ld * n), back((size_t)ld * n);
  for (size_t i = 0; i < host.size(); ++i) host[i] = (double)(i % 1000003);
  std::vector<hipStream_t> st(ns);
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<double*> dev((size_t)blocks * blocks, nullptr);
  for (auto& d : dev) CK(hipMalloc(&d, (size_t)nb * nb * sizeof(double)));
  auto block = [&](long bi, long bj, bool up) {
    const long rows = std::min(nb, n - bi * nb), cols = std::min(nb, n - bj * nb);
    double* d = dev[(size_t)(bj * blocks + bi)];
    hipStream_t s = st[(size_t)((bi + bj * blocks) % ns)];
    if (up)
      CK(hipMemcpy2DAsync(d, nb * sizeof(double), &host[(size_t)(bj * nb * ld + bi * nb)], ld * sizeof(double),
                          rows * sizeof(double), cols, hipMemcpyHostToDevice, s));
    else
      CK(hipMemcpy2DAsync(&back[(size_t)(bj * nb * ld + bi * nb)], ld * sizeof(double), d, nb * sizeof(double),
                          rows * sizeof(double), cols, hipMemcpyDeviceToHost, s));
  };
  auto check = [&](const char* what, int round) {
    long bad = 0;
    for (long j = 0; j < n; ++j)
      for (long i = 0; i < n; ++i) bad += back[(size_t)(j * ld + i)] != host[(size_t)(j * ld + i)];
    std::printf("%s round %d: %ld wrong elements%s\n", what, round, bad, failed ? " (a call failed)" : "");
    std::fflush(stdout);
    return bad == 0 && !failed;
  };
  bool ok = true;
  for (int r = 0; r < rounds && ok; ++r) {
    std::memset(back.data(), 0, back.size() * sizeof(double));
    for (long bj = 0; bj < blocks; ++bj)
      for (long bi = 0; bi < blocks; ++bi) block(bi, bj, true);
    for (long bj = 0; bj < blocks; ++bj)
      for (long bi = 0; bi < blocks; ++bi) block(bi, bj, false);
    for (auto s : st) CK(hipStreamSynchronize(s));
    ok = check("one issuing thread", r);
  }
  for (int r = 0; r < rounds && ok; ++r) {
    std::memset(back.data(), 0, back.size() * sizeof(double));
    std::vector<std::thread> th;
    for (int t = 0; t < ns; ++t)
      th.emplace_back([&, t] {
        for (int up = 1; up >= 0; --up)
          for (long b = 0; b < blocks * blocks; ++b)
            if ((int)(b % ns) == t) block(b % blocks, b / blocks, up == 1);
        CK(hipStreamSynchronize(st[(size_t)t]));
      });
    for (auto& t : th) t.join();
    ok = check("one thread per stream", r);
  }
  std::printf(ok ? "PROBE OK\n" : "PROBE FAILED\n");
  return ok ? 0 : 1;
}
