// Instantiates the MFMA GEMM kernels for one scalar kind.
#include "gemm_launch.hpp"

#include <cstdlib>
namespace eigamd {
// Two tiles for double: 256 x 128 (one workgroup per CU) and 128 x 64 (two per CU).  The large one is the faster
// per flop; the small one wins where the large one's tiles are too few or too long for the chip -- decided by the
// launcher's cost model for either (EIGAMD_F64_TILE = 1 / 2 fixes the large / small tile: tests, measurements).
hipError_t launch_gemm_f64(const GemmCall& c, hipStream_t s) {
  const char* e = std::getenv("EIGAMD_F64_TILE");  // (read per call: tests set it between calls)
  const int fixed = e ? std::atoi(e) : 0;
  if (fixed == 1) return launch_gemm<F64>(c, s);
  if (fixed == 2) return launch_gemm_f64_small(c, s);
  // A product that overwrites one of its operands (the triangular solves' B <- inv * B) relies on one large tile
  // covering the operand's whole extent: a workgroup reads all it needs before it stores.
  const bool in_place = c.C == c.A || c.C == c.B;
  if (c.m <= 0 || c.n <= 0 || c.k <= 0 || c.atomic_c || in_place) return launch_gemm<F64>(c, s);
  const int64_t large_tiles = ((c.m + Geometry<F64>::BM - 1) / Geometry<F64>::BM) * ((c.n + Geometry<F64>::BN - 1) / Geometry<F64>::BN);
  if (large_tiles >= 2 * 256) return launch_gemm<F64>(c, s);
  return modelled_cost<F64S>(c) < 0.95 * modelled_cost<F64>(c) ? launch_gemm_f64_small(c, s) : launch_gemm<F64>(c, s);
}
}
