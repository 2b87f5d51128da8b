// Plumbing of the one-process-per-GPU drivers (eigen_amd/summa.py, eigen_amd/plu.py): what a process
// needs in order to read a panel piece straight out of another process's HBM with the copy engines, and
// to keep a few compute units free of the local products.
//
// The reference has no distributed layer.  Its parallel split (parallelize_gemm,
// Eigen/src/Core/products/Parallelizer.h:101-176) lets all threads read ONE packed A panel that they built
// together (GeneralMatrixMatrix.h:100-136: `blocking.blockA()` is shared, `info[i].sync / users` say when a
// slice may be read or overwritten).  Between processes that own one GPU each, "shared" is device memory
// opened through an inter-process handle, and a read is a device-to-device copy on a copy stream: the DMA
// engines move it over the xGMI link between the two devices, no compute unit takes part.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/eigen_amd_blas.h"

namespace eigamd {
int abi_fail(int code, const std::string& what);  // blas_abi.hip: sets eigamd_last_error()
}
using namespace eigamd;

#define XF_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      (void)hipGetLastError();                                                         \
      return abi_fail(EIGAMD_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    }                                                                                  \
  } while (0)

extern "C" {

int eigamd_ipc_export(const void* d_ptr, void* handle, int64_t* offset, int64_t* span) {
  if (!d_ptr || !handle || !offset) return abi_fail(EIGAMD_ERR_ARG, "invalid ipc_export arguments");
  static_assert(sizeof(hipIpcMemHandle_t) <= EIGAMD_IPC_HANDLE_BYTES, "handle size");
  // A handle stands for a whole allocation: find the allocation `d_ptr` lies in.
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  XF_HIP(hipMemGetAddressRange(&base, &size, const_cast<void*>(d_ptr)));
  hipIpcMemHandle_t h;
  XF_HIP(hipIpcGetMemHandle(&h, base));
  std::memset(handle, 0, EIGAMD_IPC_HANDLE_BYTES);
  std::memcpy(handle, &h, sizeof h);
  *offset = static_cast<const char*>(d_ptr) - static_cast<const char*>(base);
  if (span) *span = (int64_t)size;
  return EIGAMD_OK;
}

int eigamd_ipc_open(const void* handle, void** d_base) {
  if (!handle || !d_base) return abi_fail(EIGAMD_ERR_ARG, "invalid ipc_open arguments");
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof h);
  *d_base = nullptr;
  XF_HIP(hipIpcOpenMemHandle(d_base, h, hipIpcMemLazyEnablePeerAccess));
  return EIGAMD_OK;
}

int eigamd_ipc_close(void* d_base) {
  if (!d_base) return EIGAMD_OK;
  XF_HIP(hipIpcCloseMemHandle(d_base));
  return EIGAMD_OK;
}

int eigamd_copy_dev_async(void* d_dst, const void* d_src, int64_t bytes, void* stream) {
  if (bytes < 0 || (bytes > 0 && (!d_dst || !d_src))) return abi_fail(EIGAMD_ERR_ARG, "invalid copy arguments");
  if (bytes == 0) return EIGAMD_OK;
  // hipMemcpyDefault: the runtime finds the owning devices itself (a pointer opened through an inter-process
  // handle belongs to the exporting process's device).
  XF_HIP(hipMemcpyAsync(d_dst, d_src, (size_t)bytes, hipMemcpyDefault, static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

int eigamd_stream_create_cu_mask(int leave_free, int take_free, void** stream) {
  if (!stream || leave_free < 0 || take_free < 0 || (leave_free > 0) == (take_free > 0))
    return abi_fail(EIGAMD_ERR_ARG, "exactly one of leave_free / take_free must be positive");
  *stream = nullptr;
  int dev = 0;
  XF_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  XF_HIP(hipGetDeviceProperties(&prop, dev));
  const int total = prop.multiProcessorCount;
  int n = leave_free > 0 ? leave_free : take_free;
  // The same number from every XCD and from every shader engine of it (bit b of the mask: compute unit b / 8 of XCD b % 8;
  // four engines per XCD): a launch's workgroups are dealt evenly over the engines whatever the mask says, so an engine
  // left with fewer units than the others sets the pace for all of them -- hence steps of 32.
  n = (n + 31) / 32 * 32;
  if (total % 32 != 0 || n * 2 > total) return abi_fail(EIGAMD_ERR_ARG, "cannot set that many compute units aside on this device");
  std::vector<uint32_t> mask((size_t)total / 32, 0u);
  for (int b = 0; b < total; ++b) {
    const bool top = b >= total - n;
    if (top == (take_free > 0)) mask[(size_t)b / 32] |= 1u << (b % 32);
  }
  hipStream_t s = nullptr;
  XF_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  *stream = s;
  return EIGAMD_OK;
}

int eigamd_stream_destroy(void* stream) {
  if (!stream) return EIGAMD_OK;
  XF_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  XF_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return EIGAMD_OK;
}

}  // extern "C"
