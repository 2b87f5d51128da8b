// Instantiates the MFMA GEMM kernels for one scalar kind.
#include "gemm_launch.hpp"

#include <cstdlib>
namespace eigamd {
// Two tiles: 128 x 128 (one workgroup per CU) and 64 x 64 (two per CU); see gemm_inst_f64.hip
// (EIGAMD_C64_TILE = 1 / 2 fixes the large / small tile).
hipError_t launch_gemm_c64(const GemmCall& c, hipStream_t s) {
  return small_tile_pays<C64, C64S>(c, "EIGAMD_C64_TILE") ? launch_gemm_c64_small(c, s) : launch_gemm<C64>(c, s);
}
void describe_gemm_c64(const GemmCall& c, int* out) { describe_plan<C64, C64S>(c, "EIGAMD_C64_TILE", out); }
}
