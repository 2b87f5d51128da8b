// Measures the issue-bound MFMA rate of this device for the f64 16x16x4 and f32
// 32x32x2 instructions (independent accumulators, operands in registers), i.e.
// the ceiling the GEMM kernels are priced against.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int NACC> __global__ void __launch_bounds__(256) peak_f64(double* out, int iters, double a0, double b0) {
  d4 acc[NACC]; for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; i++) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  double s = 0; for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ void __launch_bounds__(256) peak_f32(float* out, int iters, float a0, float b0) {
  f16 acc[NACC]; for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; i++) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  float s = 0; for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  void* out; hipMalloc(&out, 256 * 8 * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    int blocks = 256 * waves_per_simd;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(peak_f64<8>, dim3(blocks), dim3(256), 0, 0, (double*)out, iters, 0.5, 0.25); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = (double)blocks * 4 * iters * 8 * 2048.0;
      printf("f64 16x16x4  waves/SIMD=%d  %.3f ms  %.2f TFLOP/s\n", waves_per_simd, ms, fl / ms * 1e-9);
    }
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(peak_f32<4>, dim3(blocks), dim3(256), 0, 0, (float*)out, iters, 0.5f, 0.25f); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = (double)blocks * 4 * iters * 4 * 4096.0;
      printf("f32 32x32x2  waves/SIMD=%d  %.3f ms  %.2f TFLOP/s\n", waves_per_simd, ms, fl / ms * 1e-9);
    }
  }
  return 0;
}
