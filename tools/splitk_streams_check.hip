// Products whose reduction the launcher cuts into pieces, queued on more streams than the library has buffers for
// the pieces: the buffers must change hands (first without any event between uses of one stream, then -- from the first
// hand-over on -- fenced by events), and every product must still be the bits of the same product run alone.
//   splitk_streams_check [streams=12] [rounds=6] [n=1024]
// Linked against the toolkit's own HIP runtime, where hipStreamGetId exists (a process that imported PyTorch first may
// run an older runtime and then starts in the fenced mode).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/eigen_amd_blas.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 2; } } while (0)
#define EK(x) do { if ((x) != 0) { printf("library error at %s: %s\n", #x, eigamd_last_error()); return 2; } } while (0)

int main(int argc, char** argv) {
  const int ns = argc > 1 ? atoi(argv[1]) : 12, rounds = argc > 2 ? atoi(argv[2]) : 6;
  const int64_t n = argc > 3 ? atoll(argv[3]) : 1024;
  const size_t bytes = (size_t)n * n * sizeof(double);
  const double one = 1.0, half = 0.5;
  std::vector<hipStream_t> st(ns);
  std::vector<double*> A(ns), B(ns), C(ns);
  for (int s = 0; s < ns; ++s) {
    CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
    CK(hipMalloc(&A[s], bytes)); CK(hipMalloc(&B[s], bytes)); CK(hipMalloc(&C[s], bytes));
    EK(eigamd_fill_uniform_dev(EIGAMD_F64, A[s], n, n, n, 10 + s, nullptr));
    EK(eigamd_fill_uniform_dev(EIGAMD_F64, B[s], n, n, n, 50 + s, nullptr));
  }
  CK(hipDeviceSynchronize());
  // what every stream's product is when it runs alone
  std::vector<std::vector<double>> want(ns, std::vector<double>((size_t)n * n)), got(ns, std::vector<double>((size_t)n * n));
  for (int s = 0; s < ns; ++s) {
    EK(eigamd_fill_uniform_dev(EIGAMD_F64, C[s], n, n, n, 90 + s, nullptr));
    EK(eigamd_gemm_dev(EIGAMD_F64, EIGAMD_OP_N, EIGAMD_OP_N, n, n, n, &one, A[s], n, B[s], n, &half, C[s], n, nullptr));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(want[s].data(), C[s], bytes, hipMemcpyDeviceToHost));
  }
  int plan[5] = {0, 0, 0, 0, 0};
  EK(eigamd_gemm_plan(EIGAMD_F64, n, n, n, 0, plan));
  printf("plan for %lld^3: small tile %d, pieces %d, tail pieces %d\n", (long long)n, plan[0], plan[1], plan[2]);
  if (plan[1] < 2 && plan[2] < 2) { printf("this size is not cut into pieces: nothing to check\n"); return 3; }
  long bad = 0;
  for (int r = 0; r < rounds; ++r) {
    // all streams at once, several products in a row on each (the second overwrites the first's result: same bits expected)
    for (int rep = 0; rep < 2; ++rep)
      for (int s = 0; s < ns; ++s) {
        EK(eigamd_fill_uniform_dev(EIGAMD_F64, C[s], n, n, n, 90 + s, st[s]));
        EK(eigamd_gemm_dev(EIGAMD_F64, EIGAMD_OP_N, EIGAMD_OP_N, n, n, n, &one, A[s], n, B[s], n, &half, C[s], n, st[s]));
      }
    for (int s = 0; s < ns; ++s) CK(hipStreamSynchronize(st[s]));
    for (int s = 0; s < ns; ++s) {
      CK(hipMemcpy(got[s].data(), C[s], bytes, hipMemcpyDeviceToHost));
      if (memcmp(got[s].data(), want[s].data(), bytes) != 0) {
        ++bad;
        printf("round %d stream %d: result differs from the product run alone\n", r, s);
      }
    }
    // a stream is destroyed and a new one created: its handle may be the old one's value, its id is not
    if (r + 1 < rounds) {
      CK(hipStreamDestroy(st[r % ns]));
      CK(hipStreamCreateWithFlags(&st[r % ns], hipStreamNonBlocking));
    }
  }
  printf(bad ? "SPLITK STREAMS CHECK FAILED (%ld)\n" : "SPLITK STREAMS CHECK OK (%ld)\n", bad);
  return bad ? 1 : 0;
}
