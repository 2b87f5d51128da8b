// Read-modify-write of a column-major f64 matrix tile by tile (256 x 128 per workgroup, 64 x 64
// per wavefront), with the two lane -> element maps a GEMM epilogue can use:
//   pattern 0: the accumulator layout of v_mfma_f64_16x16x4_f64 -- one instruction touches
//              four columns, 16 consecutive rows (128 bytes) in each;
//   pattern 1: one instruction touches one column, 64 consecutive rows (512 bytes).
// Prints the time and the bandwidth of C = 2 * C.  Usage: c_pattern [n] [ld_pad]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int PATTERN>
__global__ void __launch_bounds__(512) rmw(double* C, long ld, int tiles_m) {
  const int tile_m = blockIdx.x % tiles_m, tile_n = blockIdx.x / tiles_m;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm0 = (wave % 4) * 64, wn0 = (wave / 4) * 64;
  double* base = C + (long)tile_m * 256 + wm0 + ((long)tile_n * 128 + wn0) * ld;
  if (PATTERN == 0) {
    const int xi = lane % 16, kk = lane / 16;
    for (int j = 0; j < 4; ++j) {
      double v[16];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i * 4 + r] = base[i * 16 + xi + (long)(j * 16 + 4 * r + kk) * ld];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) base[i * 16 + xi + (long)(j * 16 + 4 * r + kk) * ld] = 2.0 * v[i * 4 + r];
    }
  } else {
    for (int j = 0; j < 4; ++j) {
      double v[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) v[c] = base[lane + (long)(j * 16 + c) * ld];
#pragma unroll
      for (int c = 0; c < 16; ++c) base[lane + (long)(j * 16 + c) * ld] = 2.0 * v[c];
    }
  }
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 32768, ld = n + (argc > 2 ? atol(argv[2]) : 0);
  double* C;
  if (hipMalloc(&C, sizeof(double) * ld * n) != hipSuccess) return 1;
  hipMemset(C, 0, sizeof(double) * ld * n);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles_m = (int)(n / 256), tiles = tiles_m * (int)(n / 128);
  for (int pattern = 0; pattern < 2; ++pattern) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (pattern == 0) hipLaunchKernelGGL(rmw<0>, dim3(tiles), dim3(512), 0, 0, C, ld, tiles_m);
      else hipLaunchKernelGGL(rmw<1>, dim3(tiles), dim3(512), 0, 0, C, ld, tiles_m);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("pattern %d (%s): n=%ld ld=%ld  %.3f ms  %.2f TB/s (read + write)\n", pattern,
           pattern == 0 ? "4 columns x 128 B per instruction" : "1 column x 512 B per instruction", n, ld, best,
           2.0 * n * n * 8 / best / 1e9);
  }
  return 0;
}
