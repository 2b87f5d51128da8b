// Instantiates the MFMA GEMM kernels for one scalar kind with its small tile (64 x 128).
#include "gemm_launch.hpp"
namespace eigamd {
hipError_t launch_gemm_c32_small(const GemmCall& c, hipStream_t s) { return launch_gemm<C32S>(c, s); }
}
