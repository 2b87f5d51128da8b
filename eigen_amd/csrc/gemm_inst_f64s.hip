// Instantiates the MFMA GEMM kernels for double with the 128 x 64 tile.
#include "gemm_launch.hpp"
namespace eigamd {
hipError_t launch_gemm_f64_small(const GemmCall& c, hipStream_t s) { return launch_gemm<F64S>(c, s); }
}
