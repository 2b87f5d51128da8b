// Counter-based U[-1,1) generator shared by host and device.  Element (i, j) of
// a matrix depends only on (seed, i, j): h = mix(seed ^ mix((j << 32) | i)) with
// the SplitMix64 finaliser; the top 53 (double) or 24 (float) bits become the
// mantissa.  The imaginary part of a complex element uses mix(h + GAMMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eigamd {

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t elem_hash(uint64_t seed, int64_t i, int64_t j) {
  return mix64(seed ^ mix64(((uint64_t)j << 32) | (uint64_t)i));
}
__host__ __device__ inline double unit_f64(uint64_t h) {
  return (double)(h >> 11) * (1.0 / 4503599627370496.0) - 1.0;  // 2^-52 * [0,2^53) - 1
}
__host__ __device__ inline float unit_f32(uint64_t h) {
  return (float)(h >> 40) * (1.0f / 8388608.0f) - 1.0f;  // 2^-23 * [0,2^24) - 1
}

template <typename F> __host__ __device__ inline void put_uniform(int dtype, void* x, int64_t off, uint64_t h) {
  switch (dtype) {
    case 0: static_cast<float*>(x)[off] = unit_f32(h); break;
    case 1: static_cast<double*>(x)[off] = unit_f64(h); break;
    case 2: static_cast<float*>(x)[2 * off] = unit_f32(h);
            static_cast<float*>(x)[2 * off + 1] = unit_f32(mix64(h + 0x9E3779B97F4A7C15ull)); break;
    default: static_cast<double*>(x)[2 * off] = unit_f64(h);
             static_cast<double*>(x)[2 * off + 1] = unit_f64(mix64(h + 0x9E3779B97F4A7C15ull)); break;
  }
}

// Block-cyclic index map: local index l of process p (of np, block nb) is global
// index (l / nb) * nb * np + p * nb + l % nb.  np == 1 is the identity.
struct CyclicMap {
  int64_t nb;
  int np, p;
  __host__ __device__ int64_t global(int64_t l) const {
    return np == 1 ? l : (l / nb) * nb * np + (int64_t)p * nb + l % nb;
  }
};

__global__ void fill_uniform_kernel(int dtype, void* x, int64_t rows, int64_t cols, int64_t ld, uint64_t seed,
                                    CyclicMap rmap, CyclicMap cmap) {
  const int64_t total = rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx / rows, i = idx - j * rows;
    put_uniform<void>(dtype, x, i + j * ld, elem_hash(seed, rmap.global(i), cmap.global(j)));
  }
}

inline hipError_t fill_uniform_device(int dtype, void* x, int64_t rows, int64_t cols, int64_t ld,
                                      uint64_t seed, CyclicMap rmap, CyclicMap cmap, hipStream_t stream) {
  const int64_t total = rows * cols;
  const int64_t want = (total + 255) / 256;
  const unsigned blocks = (unsigned)(want > 16384 ? 16384 : want);
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(blocks), dim3(256), 0, stream, dtype, x, rows, cols, ld, seed,
                     rmap, cmap);
  return hipGetLastError();
}

inline void fill_uniform_host(int dtype, void* x, int64_t rows, int64_t cols, int64_t ld, uint64_t seed) {
  for (int64_t j = 0; j < cols; ++j)
    for (int64_t i = 0; i < rows; ++i) put_uniform<void>(dtype, x, i + j * ld, elem_hash(seed, i, j));
}

}  // namespace eigamd
