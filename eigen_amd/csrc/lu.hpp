// Device-resident blocked LU (lu.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eigamd {

// In-place P*A = L*U of the m x n column-major matrix at A (device memory), kind F32
// or F64.  d_transp (device, min(m,n) ints) receives the 0-based row transpositions
// as global row indices; *first_zero_pivot the index of the first exactly-zero pivot
// or -1; *nb_transpositions the number of actual row swaps.  Synchronises `s`.
hipError_t getrf_device(int kind, int64_t m, int64_t n, void* A, int64_t ld, int* d_transp, int* first_zero_pivot,
                        int* nb_transpositions, hipStream_t s);

}  // namespace eigamd
