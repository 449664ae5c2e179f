/*
 * dbcsr_acc_libsmm.h -- the small-matrix-multiply ("libsmm_acc") part of the
 * accelerator C-ABI, re-exported by libdbcsr_acc_amd.so.
 *
 * Replaces /root/reference/src/acc/acc_libsmm.h:31-49 (implemented in the
 * reference by src/acc/libsmm_acc/libsmm_acc.cpp and
 * src/acc/cuda_hip/calculate_norms.cpp); bound on the Fortran side by
 * src/mm/dbcsr_acc_operations.F:38-135 and src/mm/dbcsr_mm_common.F:117-130.
 *
 * libsmm_acc_process return convention (src/mm/dbcsr_acc_operations.F:134-135,
 * src/mm/dbcsr_mm_accdrv.F:531-538): >= 0 the stack was executed on the device
 * (10 = executed by a generic, not shape-specialised kernel); < 0 the stack was
 * NOT handled and the caller must run it on the host:
 *   -1  inhomogeneous stack (def_mnk != 1)            libsmm_acc.cpp:327
 *   -10 data type not supported (complex)             libsmm_acc.cpp:338
 *   -20 no kernel for this (m,n,k)                    libsmm_acc.cpp:305
 * Never aborts for an unsupported shape.
 *
 * Differences from the reference, by design:
 *   - dbcsr_type_real_4 stacks run on the device (fp32 MFMA kernels); the
 *     reference returns -10 for them.
 *   - blocks with a dimension > max_kernel_dim run through a tiled device
 *     kernel on stack_stream's data (the reference loops hipBLAS calls).
 */
#ifndef DBCSR_AMD_ACC_LIBSMM_H
#define DBCSR_AMD_ACC_LIBSMM_H

#include "dbcsr_acc.h"

#if defined(__cplusplus)
extern "C" {
#endif

/* acc_libsmm.h:31-36 -- DBCSR data type codes (src/data/dbcsr_data_types.F:122-133) */
typedef enum libsmm_acc_data_t {
  dbcsr_type_real_4 = 1,
  dbcsr_type_real_8 = 3,
  dbcsr_type_complex_4 = 5,
  dbcsr_type_complex_8 = 7
} libsmm_acc_data_t;

/* acc_libsmm.h:38-40 */
int libsmm_acc_init(void);
int libsmm_acc_finalize(void);
c_dbcsr_acc_bool_t libsmm_acc_is_thread_safe(void);

/* acc_libsmm.h:42-43.  In-place transpose of stack_size blocks (each m x n,
 * column-major) of dev_data; dev_trs_stack[offset + i] is the 0-based element
 * offset of block i.  No-op (0) for non-fp64 types or m,n > max_kernel_dim, as
 * the reference (libsmm_acc.cpp:482-487). */
int libsmm_acc_transpose(const int* dev_trs_stack, int offset, int stack_size, void* dev_data, libsmm_acc_data_t datatype, int m,
  int n, int max_kernel_dim, void* stream);

/* acc_libsmm.h:45-47.  Executes one parameter stack:
 *   for each entry s: C[c_s : c_s+m*n] += A[a_s : a_s+m*k] (m x k, col-major)
 *                                        * B-block b_s
 * host_param_stack: 7 x stack_size int32 (m,n,k,a,b,c,c_blk), 1-based offsets
 *                   (src/mm/dbcsr_mm_types.F:24-37); may be NULL here.
 * dev_param_stack : 3 x stack_size int32 (a,b,c), 1-based element offsets,
 *                   resident in device memory.
 * B blocks are read in the layout libsmm_acc_transpose leaves them in
 * (n x k column-major) for fp64 with all dims <= max_kernel_dim, and as stored
 * (k x n column-major) otherwise -- the same pairing as the reference.
 * Kernels run on stack_stream; c_stream may equal stack_stream. */
int libsmm_acc_process(const int* host_param_stack, const int* dev_param_stack, int stack_size, libsmm_acc_data_t datatype,
  const void* dev_a_data, const void* dev_b_data, void* dev_c_data, int m_max, int n_max, int k_max, int max_kernel_dim,
  c_dbcsr_acc_bool_t def_mnk, void* stack_stream, void* c_stream);

/* acc_libsmm.h:49.  norms[i] = sum_j mat[offsets[i] + j]^2, j < nelems[i]
 * (fp64 in, fp32 out); offsets/nelems/norms are device arrays. */
int c_calculate_norms(const double* mat, int nblks, const int* offsets, const int* nelems, float* norms, void* stream_ptr);

/* Declared by the host in src/core/dbcsr_lib.F:110-116; harmless to export. */
int libsmm_acc_gpu_warp_size(void);

#if defined(__cplusplus)
}
#endif
#endif
