// Instantiates the MFMA GEMM kernels for one scalar kind.
#include "gemm_launch.hpp"
namespace eigamd {
hipError_t launch_gemm_c32(const GemmCall& c, hipStream_t s) { return launch_gemm<C32>(c, s); }
}
