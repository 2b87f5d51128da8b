// Instantiates the MFMA GEMM kernels for one scalar kind.
#include "gemm_launch.hpp"

#include <cstdlib>
namespace eigamd {
// Two tiles for float: 256 x 256 (one workgroup per CU) and 128 x 128 (two per CU); see gemm_inst_f64.hip
// (EIGAMD_F32_TILE = 1 / 2 fixes the large / small tile).
hipError_t launch_gemm_f32(const GemmCall& c, hipStream_t s) {
  return small_tile_pays<F32, F32S>(c, "EIGAMD_F32_TILE") ? launch_gemm_f32_small(c, s) : launch_gemm<F32>(c, s);
}
void describe_gemm_f32(const GemmCall& c, int* out) { describe_plan<F32, F32S>(c, "EIGAMD_F32_TILE", out); }
}
