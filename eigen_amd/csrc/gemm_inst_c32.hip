// Instantiates the MFMA GEMM kernels for one scalar kind.
#include "gemm_launch.hpp"

#include <cstdlib>
namespace eigamd {
// Two tiles: 128 x 256 (one workgroup per CU) and 64 x 128 (two per CU); see gemm_inst_f64.hip
// (EIGAMD_C32_TILE = 1 / 2 fixes the large / small tile).
hipError_t launch_gemm_c32(const GemmCall& c, hipStream_t s) {
  return small_tile_pays<C32, C32S>(c, "EIGAMD_C32_TILE") ? launch_gemm_c32_small(c, s) : launch_gemm<C32>(c, s);
}
void describe_gemm_c32(const GemmCall& c, int* out) { describe_plan<C32, C32S>(c, "EIGAMD_C32_TILE", out); }
}
