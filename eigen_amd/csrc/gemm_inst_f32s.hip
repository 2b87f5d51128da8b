// Instantiates the MFMA GEMM kernels for float with the 128 x 128 tile.
#include "gemm_launch.hpp"
namespace eigamd {
hipError_t launch_gemm_f32_small(const GemmCall& c, hipStream_t s) { return launch_gemm<F32S>(c, s); }
}
