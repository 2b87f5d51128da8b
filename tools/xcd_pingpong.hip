// Round-trip latency of a flag exchange between two workgroups through global memory, for the
// memory scopes a pivot exchange could use.  Every wait is bounded: a scope that is not coherent
// between the two workgroups shows up as "gave up", not as a hang.  (Confining a stream to the
// CUs of one XCD, mask bits b with b % 8 == xcd, is not possible on this runtime:
// hipExtStreamCreateWithCUMask crashes on such a mask.)
//   xcd_pingpong [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SCOPE> __device__ __forceinline__ unsigned ld(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE);
}
template <int SCOPE> __device__ __forceinline__ void st(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE);
}

template <int SCOPE>
__global__ void pingpong(unsigned* flags, int rounds, int* failed, unsigned* xcc) {
  unsigned* ping = flags;        // written by workgroup 0
  unsigned* pong = flags + 256;  // written by workgroup 1 (1 KiB away)
  if (threadIdx.x != 0) return;
  unsigned id = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  xcc[blockIdx.x] = id & 0xf;
  for (int i = 1; i <= rounds; ++i) {
    if (blockIdx.x == 0) {
      st<SCOPE>(ping, (unsigned)i);
      unsigned spins = 0;
      while (ld<SCOPE>(pong) != (unsigned)i) {
        if (++spins > 2000000u || ld<__HIP_MEMORY_SCOPE_AGENT>((const unsigned*)failed) != 0) { st<__HIP_MEMORY_SCOPE_AGENT>((unsigned*)failed, 1u); return; }
      }
    } else {
      unsigned spins = 0;
      while (ld<SCOPE>(ping) != (unsigned)i) {
        if (++spins > 2000000u || ld<__HIP_MEMORY_SCOPE_AGENT>((const unsigned*)failed) != 0) { st<__HIP_MEMORY_SCOPE_AGENT>((unsigned*)failed, 1u); return; }
      }
      st<SCOPE>(pong, (unsigned)i);
    }
  }
}

template <int SCOPE> void run(const char* name, hipStream_t s, const char* where, unsigned* flags, int* failed, unsigned* xcc, int rounds) {
  (void)hipMemsetAsync(flags, 0, 4096, s);
  (void)hipMemsetAsync(failed, 0, 4, s);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  hipLaunchKernelGGL(pingpong<SCOPE>, dim3(2), dim3(64), 0, s, flags, rounds, failed, xcc);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  int f = 0; unsigned x[2] = {99, 99};
  (void)hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
  printf("%-10s %-12s workgroups on XCC %u and %u: %s  %.3f us per round trip\n", name, where, x[0], x[1], f ? "GAVE UP" : "ok", 1e3 * ms / rounds);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 9; } } while (0)
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned* flags; int* failed; unsigned* xcc;
  if (hipMalloc(&flags, 4096) != hipSuccess || hipMalloc(&failed, 4) != hipSuccess || hipMalloc(&xcc, 8) != hipSuccess) return 1;
  static hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, words = cus / 32;
  printf("%d CUs\n", cus);
  std::vector<uint32_t> all(words, 0xffffffffu);
  hipStream_t s_all = nullptr;
  printf("creating the whole-chip stream\n");
  CK(hipExtStreamCreateWithCUMask(&s_all, (uint32_t)words, all.data()));
  printf("streams ready\n");
  run<__HIP_MEMORY_SCOPE_AGENT>("agent", s_all, "whole chip", flags, failed, xcc, rounds);
  run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup", s_all, "whole chip", flags, failed, xcc, rounds);
  run<__HIP_MEMORY_SCOPE_SYSTEM>("system", s_all, "whole chip", flags, failed, xcc, rounds);
  return 0;
}
