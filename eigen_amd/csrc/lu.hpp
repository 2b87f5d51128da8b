// Device-resident blocked LU (lu.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eigamd {

// In-place P*A = L*U of the m x n column-major matrix at A (device memory), any of the four
// kinds.  d_transp (device, min(m,n) ints) receives the 0-based row transpositions
// as global row indices; *first_zero_pivot the index of the first exactly-zero pivot
// or -1; *nb_transpositions the number of actual row swaps.  Synchronises `s`.
// max_block is the maxBlockSize of blocked_lu (256 at top level, 16 for a panel).
//
// The one-launch base case of a panel needs all its workgroups on the chip together.  When the
// device is shared with other work and they are not, its workgroups give up after a bounded
// wait and the call returns kLuNotResident with the matrix partly overwritten: the caller
// restores the matrix and calls again with allow_fused = false (per-column kernels, whose
// only grid-wide synchronisation is the kernel boundary).
constexpr hipError_t kLuNotResident = hipErrorNotReady;
hipError_t getrf_device(int kind, int64_t m, int64_t n, void* A, int64_t ld, int* d_transp, int64_t max_block,
                        int* first_zero_pivot, int* nb_transpositions, hipStream_t s, bool allow_fused = true,
                        int* d_info = nullptr);
// With d_info (device memory, 3 ints) the call does not wait: behind `s`, d_info holds the first zero
// pivot or -1, the number of exchanges, and 1 where the call without it would return kLuNotResident.

// (One driver for the four kinds: the complex ones are the same kernels over a (re, im) scalar.)

// Row transpositions i <-> d_transp[i], i in [k0, k1), on ncols columns starting at A
// (row indices are rows of A).  Any kind.
hipError_t laswp_device(int kind, void* A, int64_t ld, int64_t ncols, const int* d_transp, int k0, int k1,
                        hipStream_t s);

// B (bs x ncols) <- L^-1 B, L the unit lower triangle of a bs x bs block.  Any kind.
hipError_t trsm_llnu_device(int kind, int64_t bs, int64_t ncols, const void* L, int64_t ldl, void* B, int64_t ldb,
                            hipStream_t s);

// op(A) X = B with the factors of getrf_device: LU (n x n, unit lower and upper triangle in place) and
// its transpositions (device, n ints, 0-based); X overwrites the n x nrhs matrix B.  Any kind; op is an
// Op of gemm_launch.hpp.  Row exchanges, then two triangular solves (the other way round, with the
// exchanges undone last, for T and C), as lapack/lu.cpp:67-84 does.  Asynchronous on `s`.
hipError_t getrs_device(int kind, int op, int64_t n, int64_t nrhs, const void* LU, int64_t ldlu, const int* d_transp, void* B,
                        int64_t ldb, hipStream_t s);

}  // namespace eigamd
