/* C ABI of libeigen_amd_blas.so: the MI355X (gfx950) implementation of Eigen's
 * dense matrix-matrix product path.  Plain pointers and sizes only.
 *
 * Every entry point names the reference interface (path relative to the Eigen
 * source tree) that it replaces or extends.
 */
#ifndef EIGEN_AMD_BLAS_H
#define EIGEN_AMD_BLAS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Scalar kinds (argument `dtype`). */
enum { EIGAMD_F32 = 0, EIGAMD_F64 = 1, EIGAMD_C32 = 2, EIGAMD_C64 = 3 };

/* Operand forms (arguments `opa`, `opb`).  N/T/C are the F77 'N','T','C'
 * (blas/common.h:47-50).  CONJ (conjugate without transposition) has no F77
 * spelling; Eigen reaches it from operator* and its BLAS binding materialises a
 * temporary for it (Eigen/src/Core/products/GeneralMatrixMatrix_BLAS.h:93-105). */
enum { EIGAMD_OP_N = 0, EIGAMD_OP_T = 1, EIGAMD_OP_C = 2, EIGAMD_OP_CONJ = 3 };

/* Return codes of the eigamd_* entry points. */
enum {
  EIGAMD_OK = 0,
  EIGAMD_ERR_ARG = -1,     /* invalid argument (the F77 entry points call xerbla_ instead) */
  EIGAMD_ERR_DEVICE = -2,  /* HIP runtime error; eigamd_last_error() has the text */
  EIGAMD_ERR_NO_GPU = -3,  /* no gfx950 device visible: there is no CPU fallback */
  EIGAMD_ERR_NOT_RESIDENT = -4 /* eigamd_getrf_dev only, see there: the matrix is partly overwritten */
};

/* ---- F77 BLAS ABI, host pointers --------------------------------------------
 * Replace EIGEN_BLAS_FUNC(gemm) (blas/level3_impl.h:12-76; prototypes
 * Eigen/src/misc/blas.h:347-351).  Same validation order and info codes
 * (1,2,3,4,5,8,10,13 -> xerbla_("?GEMM ", &info, 6)), same quick returns, beta == 0
 * overwrites without reading C, only C(0:m,0:n) is written.  Complex scalars are
 * passed as pointers to (re,im) pairs.  Return 0. */
int sgemm_(const char* transa, const char* transb, const int* m, const int* n, const int* k,
           const float* alpha, const float* a, const int* lda, const float* b, const int* ldb,
           const float* beta, float* c, const int* ldc);
int dgemm_(const char* transa, const char* transb, const int* m, const int* n, const int* k,
           const double* alpha, const double* a, const int* lda, const double* b, const int* ldb,
           const double* beta, double* c, const int* ldc);
int cgemm_(const char* transa, const char* transb, const int* m, const int* n, const int* k,
           const float* alpha, const float* a, const int* lda, const float* b, const int* ldb,
           const float* beta, float* c, const int* ldc);
int zgemm_(const char* transa, const char* transb, const int* m, const int* n, const int* k,
           const double* alpha, const double* a, const int* lda, const double* b, const int* ldb,
           const double* beta, double* c, const int* ldc);

/* ---- 64-bit host-pointer entry -----------------------------------------------
 * What general_matrix_matrix_product<...,ColMajor,1>::run computes
 * (Eigen/src/Core/products/GeneralMatrixMatrix.h:63-69): C = alpha*op(A)*op(B) +
 * beta*C with 64-bit extents and the CONJ operand form.  alpha/beta point to one
 * scalar of kind `dtype`.  Used by the header-level binding. */
int eigamd_gemm_host(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k,
                     const void* alpha, const void* a, int64_t lda, const void* b, int64_t ldb,
                     const void* beta, void* c, int64_t ldc);

/* ---- device-pointer entry ----------------------------------------------------
 * Same operation on matrices already resident in HBM (hipMalloc memory of the
 * current device).  Asynchronous on `stream` (a hipStream_t, NULL = default
 * stream).  Element alignment is enough; 16-byte aligned bases and leading
 * dimensions take the vector-load kernels. */
int eigamd_gemm_dev(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k,
                    const void* alpha, const void* d_a, int64_t lda, const void* d_b,
                    int64_t ldb, const void* beta, void* d_c, int64_t ldc, void* stream);

/* ---- the same product for callers that cannot allocate ---------------------------
 * No counterpart in the reference.  eigamd_gemm_dev may call hipMalloc on its way (the
 * partial-product buffer of a reduction split over several workgroups per tile, helper
 * streams); code under HIP graph capture, tight serving loops and harnesses that forbid
 * allocation call this variant instead: same arguments, same result bit for bit, and no
 * hipMalloc / hipFree / stream or event creation on any call.  The partial products live in a
 * fixed 128 MiB device array inside the library (one copy per GPU, held until the process
 * ends).  Limit: m * n * sizeof(scalar) <= 8 MiB (the products whose reduction the library
 * splits); a larger C is rejected with EIGAMD_ERR_ARG and eigamd_last_error() names the limit.
 * Calls whose work overlaps on one GPU (two streams, two threads) share that array and race:
 * issue them on one stream, or serialise them. */
int eigamd_gemm_dev_noalloc(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k,
                            const void* alpha, const void* d_a, int64_t lda, const void* d_b,
                            int64_t ldb, const void* beta, void* d_c, int64_t ldc, void* stream);

/* ---- the forms Eigen keeps off the F77 route ------------------------------------
 * general_matrix_matrix_product (Eigen/src/Core/products/GeneralMatrixMatrix.h:52-69) is
 * also instantiated with one real and one complex operand (gebp_traits<complex,real> /
 * <real,complex>, GeneralBlockPanelKernel.h:548-681, :925-1049) and with a destination
 * whose rows lie resIncr elements apart (ResInnerStride; blas_data_mapper with an inner
 * stride, util/BlasUtil.h).  No F77 routine expresses either, so Eigen's BLAS binding
 * leaves them on the built-in kernel (doc/UsingBlasLapackBackends.dox:59-61); the
 * header-level binding forwards them here.
 *
 * `dtype` is the kind of C, alpha and beta.  a_is_real / b_is_real: that operand holds
 * real scalars of the matching precision (dtype must then be a complex kind, and at most
 * one flag may be set).  C(i,j) is at c[i*c_inc + j*ldc] (c_inc >= 1, columns must not
 * overlap).  eigamd_gemm_mixed_dev is the mixed product on matrices already in HBM. */
int eigamd_gemm_ex_host(int dtype, int a_is_real, int b_is_real, int opa, int opb, int64_t m, int64_t n, int64_t k,
                        const void* alpha, const void* a, int64_t lda, const void* b, int64_t ldb, const void* beta,
                        void* c, int64_t c_inc, int64_t ldc);
int eigamd_gemm_mixed_dev(int dtype, int a_is_real, int b_is_real, int opa, int opb, int64_t m, int64_t n, int64_t k,
                          const void* alpha, const void* d_a, int64_t lda, const void* d_b, int64_t ldb, const void* beta,
                          void* d_c, int64_t ldc, void* stream);

/* ---- the product over several GPUs of one node ---------------------------------------
 * In the reference the split of a product over the cores is invisible to the caller
 * (parallelize_gemm, Eigen/src/Core/products/Parallelizer.h:101-176, reached from
 * GeneralMatrixMatrix.h:510-511): every thread owns a column slab of C, all share one packed
 * panel of A per k-block (GeneralMatrixMatrix.h:86-158).  Here one process drives an
 * nprow x npcol grid of devices: A, B and C are dealt out 2-D block-cyclically (block nb), the
 * reduction runs over k-panels of nb, and per panel the owners' pieces of A travel along grid
 * rows and those of B along grid columns, one panel ahead of the local products (SUMMA; K is
 * never split, so nothing is reduced).
 *
 * eigamd_pgemm_host: C <- alpha op(A) op(B) + beta C on host matrices (column-major, any
 * leading dimensions), scattered to the grid, multiplied, gathered.  `devices`: nprow * npcol
 * device ordinals in row-major grid order, or NULL for 0, 1, ...; an ordinal may repeat
 * ("virtual grid": several ranks on one device, for checking the schedule on one GPU).
 * `transport`: 0 = one peer copy per receiver (hipMemcpyPeerAsync, DMA engines over the xGMI
 * link of each pair), 1 = ncclBroadcast on row / column communicators (RCCL; one device per rank).
 *
 * The handle form keeps the distributed operands in HBM between calls (what a benchmark times):
 * create, fill with the synthetic operands (equal, element for element, to
 * eigamd_fill_uniform_dev of the stored matrix) or scatter / gather host matrices (`which`:
 * 0 = A, 1 = B, 2 = C, each in the orientation it is stored in), run (asynchronous), sync.
 * eigamd_pgemm_local exposes a rank's local part for inspection. */
typedef struct eigamd_pgemm eigamd_pgemm;
int eigamd_pgemm_host(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k, const void* alpha, const void* a,
                      int64_t lda, const void* b, int64_t ldb, const void* beta, void* c, int64_t ldc, int nprow, int npcol,
                      int64_t nb, const int* devices, int transport);
int eigamd_pgemm_create(int dtype, int opa, int opb, int64_t m, int64_t n, int64_t k, int nprow, int npcol, int64_t nb,
                        const int* devices, int transport, eigamd_pgemm** handle);
int eigamd_pgemm_fill(eigamd_pgemm* handle, int which, uint64_t seed);
int eigamd_pgemm_scatter(eigamd_pgemm* handle, int which, const void* host, int64_t ld);
int eigamd_pgemm_gather(eigamd_pgemm* handle, int which, void* host, int64_t ld);
int eigamd_pgemm_run(eigamd_pgemm* handle, const void* alpha, const void* beta);
int eigamd_pgemm_sync(eigamd_pgemm* handle);
int eigamd_pgemm_local(eigamd_pgemm* handle, int which, int rank, void** ptr, int64_t* rows, int64_t* cols, int64_t* ld,
                       int* device);
int eigamd_pgemm_destroy(eigamd_pgemm* handle);

/* ---- one process per GPU: reading a panel piece out of another process's HBM ----------------
 * The threads of the reference's parallel product all read ONE packed panel of A
 * (Eigen/src/Core/products/GeneralMatrixMatrix.h:100-136: blocking.blockA() is shared,
 * GemmParallelInfo::sync / users say when a slice may be read, Parallelizer.h:81-99).  Between
 * processes that own one GPU each, "shared" is device memory opened through an inter-process handle
 * and a read is a device-to-device copy on a copy stream (DMA engines over the xGMI link of the two
 * devices, no compute unit).  Used by the one-process-per-GPU drivers (eigen_amd/summa.py, plu.py).
 *
 * eigamd_ipc_export: handle (EIGAMD_IPC_HANDLE_BYTES bytes) of the allocation `d_ptr` lies in, the
 * offset of `d_ptr` inside it and the allocation's size; eigamd_ipc_open (in another process): the
 * allocation's base address in this process; eigamd_ipc_close undoes it.  eigamd_copy_dev_async:
 * `bytes` bytes between any two device addresses valid in this process, asynchronous on `stream`.
 *
 * eigamd_stream_create_cu_mask: a stream whose kernels run on a subset of the compute units, the same
 * number from every XCD: with leave_free = n on all but n of them, with take_free = n on exactly those
 * n (rounded up to a multiple of 32: one unit per shader engine of every XCD).  The local products of the RCCL transport run on the first kind,
 * so that the compute units a broadcast kernel needs are never all held by product workgroups. */
#define EIGAMD_IPC_HANDLE_BYTES 64
int eigamd_ipc_export(const void* d_ptr, void* handle, int64_t* offset, int64_t* span);
int eigamd_ipc_open(const void* handle, void** d_base);
int eigamd_ipc_close(void* d_base);
int eigamd_copy_dev_async(void* d_dst, const void* d_src, int64_t bytes, void* stream);
int eigamd_stream_create_cu_mask(int leave_free, int take_free, void** stream);
int eigamd_stream_destroy(void* stream);

/* ---- synthetic operands ------------------------------------------------------
 * U[-1,1) entries from a counter-based generator: element (i,j) depends only on
 * (seed, i, j), so host and device produce bit-identical matrices without a
 * transfer.  Mirrors the operand set-up of bench/bench_gemm.cpp:243-246
 * (setRandom), whose std::rand stream is libc-dependent. */
int eigamd_fill_uniform_dev(int dtype, void* d_x, int64_t rows, int64_t cols, int64_t ld,
                            uint64_t seed, void* stream);
int eigamd_fill_uniform_host(int dtype, void* x, int64_t rows, int64_t cols, int64_t ld,
                             uint64_t seed);
/* The local part of a matrix distributed 2-D block-cyclically (block nb_r x nb_c
 * over an nprow x npcol process grid, this process at (myrow, mycol)): local
 * element (i, j) gets the value of its GLOBAL position, so the distributed matrix
 * equals the one eigamd_fill_uniform_dev writes on one device.  No counterpart in
 * the reference (it has no distributed layer); used by the multi-GPU GEMM driver. */
int eigamd_fill_uniform_cyclic_dev(int dtype, void* d_x, int64_t rows, int64_t cols, int64_t ld,
                                   uint64_t seed, int64_t nb_r, int nprow, int myrow, int64_t nb_c,
                                   int npcol, int mycol, void* stream);

/* ---- blocked LU with partial pivoting ---------------------------------------------
 * The trailing update A22 -= A21*A12 of the blocked factorisation is the GEMM path.
 *
 * eigamd_blocked_lu_host replaces partial_lu_impl<Scalar,ColMajor,int>::blocked_lu
 * (Eigen/src/LU/PartialPivLU.h:429-498; the seam Eigen/src/LU/PartialPivLU_LAPACKE.h:46-66
 * uses): in-place factors, 0-based row transpositions (row i was swapped with row
 * row_transpositions[i]), the number of actual swaps, and through *first_zero_pivot
 * the index of the first exactly-zero pivot or -1.  Same block-size rule, recursion
 * and pivot choice as the reference, so pivots match Eigen's.
 *
 * sgetrf_/dgetrf_/cgetrf_/zgetrf_ replace EIGEN_LAPACK_FUNC(getrf) (lapack/lu.cpp:14-41): 1-based
 * ipiv, info = -i for a bad argument i (through xerbla_), else first zero pivot + 1.
 *
 * eigamd_getrf_dev factorises a matrix already in HBM; d_transp is device memory
 * (min(m,n) ints); max_block_size is blocked_lu's maxBlockSize argument (256 for a
 * whole matrix, 16 for one panel, PartialPivLU.h:429,472).  It synchronises `stream`
 * before returning the two counters.
 *
 * Sharing the device: the kernel that factorises 16 columns of a tall panel in one launch needs all its
 * workgroups on the device at the same time.  When kernels of other streams or processes hold the compute
 * units, its workgroups give up after a bounded wait (about a second in all; nothing hangs) and
 * eigamd_getrf_dev returns EIGAMD_ERR_NOT_RESIDENT.  d_a and d_transp then hold a factorisation that stopped
 * part-way: columns left of the panel that gave up are final, the rest is neither the input nor the result.
 * The caller restores d_a from its copy and calls eigamd_getrf_dev_shared, which factorises the same way
 * (same pivots, same factors) with one launch per pivot column in place of that kernel, needs no residency
 * and therefore never returns this code; it is the slower form (about 2x at 8192 x 8192) and the right first
 * call for a process that knows it shares its GPU.  The host-pointer entries above (?getrf_,
 * eigamd_blocked_lu_host*) still hold the caller's matrix and do exactly this by themselves. */
int eigamd_blocked_lu_host(int dtype, int64_t rows, int64_t cols, void* lu, int64_t lu_stride,
                           int* row_transpositions, int* nb_transpositions, int* first_zero_pivot);
/* The same for a matrix stored row by row (row stride lu_stride >= cols): what
 * partial_lu_impl<Scalar,RowMajor,int>::blocked_lu factorises (the reference instantiates its blocked_lu for
 * either storage order, PartialPivLU.h:333-341; its LAPACKE binding passes the order on,
 * PartialPivLU_LAPACKE.h:50-64).  Same factors, transpositions and counters.
 *
 * Complex kinds: the reference's blocked_lu is a template and its bindings include cgetrf / zgetrf
 * (PartialPivLU_LAPACKE.h:77-78, lapack/complex_single.cpp, complex_double.cpp).  All the LU entry points
 * here take the four kinds; the pivot of a complex column is its first entry of largest modulus
 * (scalar_score_coeff_op, PartialPivLU.h:375).  One driver serves the four kinds. */
int eigamd_blocked_lu_host_rowmajor(int dtype, int64_t rows, int64_t cols, void* lu, int64_t lu_stride,
                                    int* row_transpositions, int* nb_transpositions, int* first_zero_pivot);
int dgetrf_(const int* m, const int* n, double* a, const int* lda, int* ipiv, int* info);
int sgetrf_(const int* m, const int* n, float* a, const int* lda, int* ipiv, int* info);
int cgetrf_(const int* m, const int* n, void* a, const int* lda, int* ipiv, int* info);   /* a: interleaved (re, im) floats */
int zgetrf_(const int* m, const int* n, void* a, const int* lda, int* ipiv, int* info);   /* a: interleaved (re, im) doubles */
int eigamd_getrf_dev(int dtype, int64_t m, int64_t n, void* d_a, int64_t lda, int* d_transp,
                     int64_t max_block_size, int* first_zero_pivot, int* nb_transpositions, void* stream);
int eigamd_getrf_dev_shared(int dtype, int64_t m, int64_t n, void* d_a, int64_t lda, int* d_transp,
                            int64_t max_block_size, int* first_zero_pivot, int* nb_transpositions, void* stream);
/* The same without waiting (m, n >= 1): behind `stream`, the 3 ints at d_info (device memory) hold the first
 * zero pivot or -1, the number of row exchanges, and a status (0, or 1 where eigamd_getrf_dev would have
 * returned EIGAMD_ERR_NOT_RESIDENT).  For
 * drivers that keep the whole factorisation on the device (the multi-GPU LU). */
int eigamd_getrf_dev_async(int dtype, int64_t m, int64_t n, void* d_a, int64_t lda, int* d_transp, int64_t max_block_size,
                           int* d_info, void* stream);

/* The two other steps of one blocked_lu iteration, on device memory, for drivers that
 * own the loop themselves (the multi-GPU LU): the row transpositions of
 * PartialPivLU.h:479-489 -- rows i and d_transp[i] of `ncols` columns starting at d_a
 * are exchanged for i = k0 .. k1-1, in order -- and A12 = A11^-1 A12 with A11 unit
 * lower triangular (PartialPivLU.h:492), i.e. ?trsm_("L","L","N","U", m, n, 1, ...) of
 * blas/level3_impl.h:78-178 restricted to that one form.  Any of the four kinds. */
int eigamd_laswp_dev(int dtype, void* d_a, int64_t lda, int64_t ncols, const int* d_transp, int k0, int k1,
                     void* stream);
int eigamd_trsm_llnu_dev(int dtype, int64_t m, int64_t n, const void* d_l, int64_t ldl, void* d_b, int64_t ldb,
                         void* stream);

/* ---- the solve that consumes the factors -------------------------------------------
 *
 * sgetrs_/dgetrs_/cgetrs_/zgetrs_ replace EIGEN_LAPACK_FUNC(getrs) (lapack/lu.cpp:46-89), which is
 * PartialPivLU::_solve_impl / _solve_impl_transposed (Eigen/src/LU/PartialPivLU.h:227-270) behind the
 * LAPACK interface: op(A) X = B for the n x n matrix whose ?getrf_ factors are in `a` with the 1-based
 * ipiv, X overwriting the n x nrhs matrix B.  trans 'N': B = P B, unit-lower solve, upper solve
 * (lu.cpp:67-72); 'T' / 'C': upper, then unit lower, with the transposed / adjoint triangles, then
 * B = P^T B (:73-84).  info = -i for a bad argument i through xerbla_ (trans 1, n 2, nrhs 3, lda 5,
 * ldb 8; :48-58); ipiv is left as it was (the reference shifts it to 0-based and back, :65-66, :85-86).
 * The row exchanges run 256 at a time, each run as one permutation; the two solves are the ?trsm_ of
 * this library (all four kinds).
 *
 * eigamd_getrs_dev is the same on device memory: d_lu (n x n, ld ldlu) and d_transp (n ints, 0-based, as
 * eigamd_getrf_dev leaves them), d_b n x nrhs; op is EIGAMD_OP_N / _T / _C.  Asynchronous on `stream`. */
int sgetrs_(const char* trans, const int* n, const int* nrhs, const float* a, const int* lda, int* ipiv, float* b,
            const int* ldb, int* info);
int dgetrs_(const char* trans, const int* n, const int* nrhs, const double* a, const int* lda, int* ipiv, double* b,
            const int* ldb, int* info);
int cgetrs_(const char* trans, const int* n, const int* nrhs, const void* a, const int* lda, int* ipiv, void* b,
            const int* ldb, int* info);   /* a, b: interleaved (re, im) floats */
int zgetrs_(const char* trans, const int* n, const int* nrhs, const void* a, const int* lda, int* ipiv, void* b,
            const int* ldb, int* info);   /* a, b: interleaved (re, im) doubles */
int eigamd_getrs_dev(int dtype, int op, int64_t n, int64_t nrhs, const void* d_lu, int64_t ldlu, const int* d_transp,
                     void* d_b, int64_t ldb, void* stream);

/* ---- triangular solve with many right-hand sides -------------------------------
 *
 * strsm_/dtrsm_/ctrsm_/ztrsm_ replace EIGEN_BLAS_FUNC(trsm) (blas/level3_impl.h:78-178;
 * prototypes Eigen/src/misc/blas.h): op(A) X = alpha B (side 'L') or X op(A) = alpha B
 * (side 'R'), X overwriting the m x n matrix B; A triangular, only the uplo triangle is
 * read and not the diagonal when diag = 'U'.  Same checks and xerbla_ codes (1 side,
 * 2 uplo, 3 transa, 4 diag, 5 m, 6 n, 9 lda, 11 ldb), same quick return for m == 0 or
 * n == 0; as in the reference the solve comes first and alpha is applied to the result.
 * They are what Eigen's TriangularSolverMatrix_BLAS.h:43-101 binds under EIGEN_USE_BLAS.
 *
 * eigamd_trsm_dev: the same on matrices already in HBM, 64-bit extents, asynchronous on
 * `stream`. */
#define EIGAMD_SIDE_LEFT 0
#define EIGAMD_SIDE_RIGHT 1
#define EIGAMD_UPLO_UPPER 0
#define EIGAMD_UPLO_LOWER 1
#define EIGAMD_DIAG_NONUNIT 0
#define EIGAMD_DIAG_UNIT 1
int strsm_(const char* side, const char* uplo, const char* transa, const char* diag, const int* m, const int* n,
           const float* alpha, const float* a, const int* lda, float* b, const int* ldb);
int dtrsm_(const char* side, const char* uplo, const char* transa, const char* diag, const int* m, const int* n,
           const double* alpha, const double* a, const int* lda, double* b, const int* ldb);
int ctrsm_(const char* side, const char* uplo, const char* transa, const char* diag, const int* m, const int* n,
           const float* alpha, const float* a, const int* lda, float* b, const int* ldb);
int ztrsm_(const char* side, const char* uplo, const char* transa, const char* diag, const int* m, const int* n,
           const double* alpha, const double* a, const int* lda, double* b, const int* ldb);
int eigamd_trsm_dev(int dtype, int side, int uplo, int op, int diag, int64_t m, int64_t n, const void* alpha,
                    const void* d_a, int64_t lda, void* d_b, int64_t ldb, void* stream);
/* The same on host matrices with 64-bit extents (what triangular_solve_matrix<...,ColMajor,1>::run receives,
 * Eigen/src/Core/products/TriangularSolverMatrix_BLAS.h:43-101); used by the header-level binding. */
int eigamd_trsm_host(int dtype, int side, int uplo, int op, int diag, int64_t m, int64_t n, const void* alpha,
                     const void* a, int64_t lda, void* b, int64_t ldb);

/* ---- symmetric rank-k update ----------------------------------------------------
 *
 * ssyrk_/dsyrk_/csyrk_/zsyrk_ replace EIGEN_BLAS_FUNC(syrk) (blas/level3_impl.h:357-433):
 * C <- alpha A A^T + beta C (trans 'N', A n x k) or alpha A^T A + beta C (trans 'T'; 'C' too
 * for the real kinds, an error for the complex ones as in the reference) on the uplo
 * triangle of the n x n matrix C; the other triangle is neither read nor written.  xerbla_
 * codes 1 uplo, 2 trans, 3 n, 4 k, 7 lda, 10 ldc.  {s,d}syrk_ are what Eigen's
 * GeneralMatrixMatrixTriangular_BLAS.h:49-147 binds under EIGEN_USE_BLAS.
 *
 * eigamd_syrk_dev: the same on matrices already in HBM (op is EIGAMD_OP_N or EIGAMD_OP_T),
 * asynchronous on `stream`. */
int ssyrk_(const char* uplo, const char* trans, const int* n, const int* k, const float* alpha, const float* a,
           const int* lda, const float* beta, float* c, const int* ldc);
int dsyrk_(const char* uplo, const char* trans, const int* n, const int* k, const double* alpha, const double* a,
           const int* lda, const double* beta, double* c, const int* ldc);
int csyrk_(const char* uplo, const char* trans, const int* n, const int* k, const float* alpha, const float* a,
           const int* lda, const float* beta, float* c, const int* ldc);
int zsyrk_(const char* uplo, const char* trans, const int* n, const int* k, const double* alpha, const double* a,
           const int* lda, const double* beta, double* c, const int* ldc);
int eigamd_syrk_dev(int dtype, int uplo, int op, int64_t n, int64_t k, const void* alpha, const void* d_a, int64_t lda,
                    const void* beta, void* d_c, int64_t ldc, void* stream);

/* ---- Hermitian rank-k update -----------------------------------------------------
 *
 * cherk_/zherk_ replace EIGEN_BLAS_FUNC(herk) (blas/level3_impl.h:562-624): C <- alpha A A^H +
 * beta C (trans 'N') or alpha A^H A + beta C (trans 'C'; 'T' is an error) on the uplo triangle of
 * the n x n matrix C, alpha and beta REAL (pointers to one float / double), the imaginary part of
 * the diagonal set to zero; when there is no product (k == 0 or alpha == 0) and beta == 1 nothing
 * is touched.  xerbla_ codes as ?syrk_.  (Eigen's own header binding for it is commented out,
 * GeneralMatrixMatrixTriangular_BLAS.h:134-140: reached through the F77 ABI only.)
 * eigamd_herk_dev: the same on matrices already in HBM (op is EIGAMD_OP_N or EIGAMD_OP_C). */
int cherk_(const char* uplo, const char* trans, const int* n, const int* k, const float* alpha, const float* a,
           const int* lda, const float* beta, float* c, const int* ldc);
int zherk_(const char* uplo, const char* trans, const int* n, const int* k, const double* alpha, const double* a,
           const int* lda, const double* beta, double* c, const int* ldc);
int eigamd_herk_dev(int dtype, int uplo, int op, int64_t n, int64_t k, const void* alpha, const void* d_a, int64_t lda,
                    const void* beta, void* d_c, int64_t ldc, void* stream);

/* ---- rank-2k updates ----------------------------------------------------------------
 *
 * {s,d,c,z}syr2k_ replace EIGEN_BLAS_FUNC(syr2k) (blas/level3_impl.h:435-501): C <- alpha A B^T +
 * alpha B A^T + beta C (trans 'N', A and B n x k) or alpha A^T B + alpha B^T A + beta C (trans 'T';
 * 'C' too for the real kinds, an error for the complex ones) on the uplo triangle of C.
 * cher2k_/zher2k_ replace EIGEN_BLAS_FUNC(her2k) (:626-700): alpha A B^H + conj(alpha) B A^H + beta C
 * (trans 'N') or alpha A^H B + conj(alpha) B^H A + beta C (trans 'C'), beta REAL (a pointer to one
 * float / double), the imaginary part of the diagonal zero on return.  xerbla_ codes 1 uplo, 2 trans,
 * 3 n, 4 k, 7 lda, 9 ldb, 12 ldc; return values as the reference's (?syr2k_: 1 when k == 0, else 0;
 * ?her2k_: 1).  Not bound by Eigen's headers: reached through the F77 ABI only.
 * eigamd_syr2k_dev: both on matrices already in HBM; alpha and beta point to scalars of kind `dtype`
 * (the imaginary part of beta must be zero when `hermitian`). */
int ssyr2k_(const char* uplo, const char* trans, const int* n, const int* k, const float* alpha, const float* a, const int* lda,
            const float* b, const int* ldb, const float* beta, float* c, const int* ldc);
int dsyr2k_(const char* uplo, const char* trans, const int* n, const int* k, const double* alpha, const double* a,
            const int* lda, const double* b, const int* ldb, const double* beta, double* c, const int* ldc);
int csyr2k_(const char* uplo, const char* trans, const int* n, const int* k, const float* alpha, const float* a, const int* lda,
            const float* b, const int* ldb, const float* beta, float* c, const int* ldc);
int zsyr2k_(const char* uplo, const char* trans, const int* n, const int* k, const double* alpha, const double* a,
            const int* lda, const double* b, const int* ldb, const double* beta, double* c, const int* ldc);
int cher2k_(const char* uplo, const char* trans, const int* n, const int* k, const float* alpha, const float* a, const int* lda,
            const float* b, const int* ldb, const float* beta, float* c, const int* ldc);
int zher2k_(const char* uplo, const char* trans, const int* n, const int* k, const double* alpha, const double* a,
            const int* lda, const double* b, const int* ldb, const double* beta, double* c, const int* ldc);
int eigamd_syr2k_dev(int dtype, int uplo, int op, int hermitian, int64_t n, int64_t k, const void* alpha, const void* d_a,
                     int64_t lda, const void* d_b, int64_t ldb, const void* beta, void* d_c, int64_t ldc, void* stream);

/* ---- products with a symmetric / Hermitian / triangular operand ------------------
 *
 * {s,d,c,z}symm_ and {c,z}hemm_ replace EIGEN_BLAS_FUNC(symm) / (hemm)
 * (blas/level3_impl.h:245-353, :505-560): C <- alpha A B + beta C (side 'L') or
 * alpha B A + beta C (side 'R'), A symmetric (Hermitian for ?hemm_: the imaginary part of its
 * diagonal is ignored), only the uplo triangle of A is read.  xerbla_ codes 1 side, 2 uplo,
 * 3 m, 4 n, 7 lda, 9 ldb, 12 ldc.
 * {s,d,c,z}trmm_ replace EIGEN_BLAS_FUNC(trmm) (:183-243): B <- alpha op(A) B (side 'L') or
 * alpha B op(A) (side 'R'), A triangular; codes as ?trsm_.
 * Under EIGEN_USE_BLAS they are what SelfadjointMatrixMatrix_BLAS.h and
 * TriangularMatrixMatrix_BLAS.h bind.  The *_dev forms take matrices already in HBM. */
int ssymm_(const char* side, const char* uplo, const int* m, const int* n, const float* alpha, const float* a, const int* lda,
           const float* b, const int* ldb, const float* beta, float* c, const int* ldc);
int dsymm_(const char* side, const char* uplo, const int* m, const int* n, const double* alpha, const double* a,
           const int* lda, const double* b, const int* ldb, const double* beta, double* c, const int* ldc);
int csymm_(const char* side, const char* uplo, const int* m, const int* n, const float* alpha, const float* a, const int* lda,
           const float* b, const int* ldb, const float* beta, float* c, const int* ldc);
int zsymm_(const char* side, const char* uplo, const int* m, const int* n, const double* alpha, const double* a,
           const int* lda, const double* b, const int* ldb, const double* beta, double* c, const int* ldc);
int chemm_(const char* side, const char* uplo, const int* m, const int* n, const float* alpha, const float* a, const int* lda,
           const float* b, const int* ldb, const float* beta, float* c, const int* ldc);
int zhemm_(const char* side, const char* uplo, const int* m, const int* n, const double* alpha, const double* a,
           const int* lda, const double* b, const int* ldb, const double* beta, double* c, const int* ldc);
int strmm_(const char* side, const char* uplo, const char* transa, const char* diag, const int* m, const int* n,
           const float* alpha, const float* a, const int* lda, float* b, const int* ldb);
int dtrmm_(const char* side, const char* uplo, const char* transa, const char* diag, const int* m, const int* n,
           const double* alpha, const double* a, const int* lda, double* b, const int* ldb);
int ctrmm_(const char* side, const char* uplo, const char* transa, const char* diag, const int* m, const int* n,
           const float* alpha, const float* a, const int* lda, float* b, const int* ldb);
int ztrmm_(const char* side, const char* uplo, const char* transa, const char* diag, const int* m, const int* n,
           const double* alpha, const double* a, const int* lda, double* b, const int* ldb);
int eigamd_symm_dev(int dtype, int side, int uplo, int hermitian, int64_t m, int64_t n, const void* alpha, const void* d_a,
                    int64_t lda, const void* d_b, int64_t ldb, const void* beta, void* d_c, int64_t ldc, void* stream);
int eigamd_trmm_dev(int dtype, int side, int uplo, int op, int diag, int64_t m, int64_t n, const void* alpha,
                    const void* d_a, int64_t lda, void* d_b, int64_t ldb, void* stream);

/* ---- misc --------------------------------------------------------------------- */
const char* eigamd_last_error(void);  /* text of the last failure on this thread */
int eigamd_device_ok(void);           /* 1 if a gfx950 device is usable, else 0 */
/* Block-tile extents (m, n, k) the kernels use for `dtype` (the large tile). */
void eigamd_tile_shape(int dtype, int* bm, int* bn, int* bk);
/* How eigamd_gemm_dev would launch C += A*B of these extents (column-major, no transposition), decided on the host
 * alone -- no GPU needed, nothing is launched: plan[0] = 1 when the kind's small tile is taken, plan[1] = pieces the
 * reduction is cut into for every tile (1: whole), plan[2] = pieces of the reduction of the leftover tile lines when
 * the product runs as whole rounds of tiles plus a tail (0: not that form), plan[3], plan[4] = the tile's extents
 * along m and n.  c_is_an_operand: the product overwrites one of its operands (C == B).  No counterpart in the
 * reference; for tests and for reading a profile. */
int eigamd_gemm_plan(int dtype, int64_t m, int64_t n, int64_t k, int c_is_an_operand, int* plan);

#ifdef __cplusplus
}
#endif
#endif /* EIGEN_AMD_BLAS_H */
