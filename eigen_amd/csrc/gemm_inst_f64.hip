// Instantiates the MFMA GEMM kernels for one scalar kind.
#include "gemm_launch.hpp"

#include <cstdlib>
namespace eigamd {
// Two tiles for double: 256 x 128 (one workgroup per CU) and 128 x 64 (two per CU).  The large one is the faster
// per flop; the small one wins where the large one's tiles are too few or too long for the chip -- decided by the
// launcher's cost model for either (EIGAMD_F64_TILE = 1 / 2 fixes the large / small tile).
hipError_t launch_gemm_f64(const GemmCall& c, hipStream_t s) {
  return small_tile_pays<F64, F64S>(c, "EIGAMD_F64_TILE") ? launch_gemm_f64_small(c, s) : launch_gemm<F64>(c, s);
}
void describe_gemm_f64(const GemmCall& c, int* out) { describe_plan<F64, F64S>(c, "EIGAMD_F64_TILE", out); }
}
