/*
 * dbcsr_acc.h -- the accelerator C-ABI that DBCSR's Fortran host binds to,
 * re-exported by libdbcsr_acc_amd.so (HIP on gfx950 only; no dual backend).
 *
 * Every entry point keeps the NAME, ARGUMENT MEANING and RETURN CONVENTION of
 * the interface it replaces: /root/reference/src/acc/acc.h:31-74.  The Fortran
 * side that calls them is src/acc/dbcsr_acc_{init,device,stream,event,devmem,
 * hostmem}.F (ISO_C_BINDING interfaces, unchanged).
 *
 * Conventions (behavioural spec: reference tests/dbcsr_acc_test.c):
 *   - return 0 on success, non-zero on failure (the Fortran host aborts);
 *   - a stream / event handle is an opaque void* owned by this library and
 *     released by the caller through *_destroy (NULL is legal);
 *     internally it points to a heap-allocated hipStream_t / hipEvent_t, the
 *     convention of src/acc/cuda_hip/acc_stream.cpp:40-45 that
 *     libsmm_acc_process relies on (it receives the same handles);
 *   - get_ndevices / set_active_device are legal before init;
 *   - a never-recorded event queries as "occurred";
 *   - all sizes are size_t bytes.
 */
#ifndef DBCSR_AMD_ACC_H
#define DBCSR_AMD_ACC_H

#include <stddef.h>

#if defined(__cplusplus)
extern "C" {
#endif

typedef int c_dbcsr_acc_bool_t; /* acc.h:31 */

/* acc.h:34-35 */
int c_dbcsr_acc_init(void);
int c_dbcsr_acc_finalize(void);
/* acc.h:38 */
void c_dbcsr_acc_clear_errors(void);
/* acc.h:41-43 */
int c_dbcsr_acc_get_ndevices(int* ndevices);
int c_dbcsr_acc_set_active_device(int device_id);
int c_dbcsr_acc_device_synchronize(void);
/* acc.h:46-53 -- lower priority number = higher priority */
int c_dbcsr_acc_stream_priority_range(int* least, int* greatest);
int c_dbcsr_acc_stream_create(void** stream_p, const char* name, int priority);
int c_dbcsr_acc_stream_destroy(void* stream);
int c_dbcsr_acc_stream_sync(void* stream);
int c_dbcsr_acc_stream_wait_event(void* stream, void* event);
/* acc.h:56-60 */
int c_dbcsr_acc_event_create(void** event_p);
int c_dbcsr_acc_event_destroy(void* event);
int c_dbcsr_acc_event_record(void* event, void* stream);
int c_dbcsr_acc_event_query(void* event, c_dbcsr_acc_bool_t* has_occurred);
int c_dbcsr_acc_event_synchronize(void* event);
/* acc.h:63-73 */
int c_dbcsr_acc_dev_mem_allocate(void** dev_mem, size_t nbytes);
int c_dbcsr_acc_dev_mem_deallocate(void* dev_mem);
int c_dbcsr_acc_dev_mem_set_ptr(void** dev_mem, void* other, size_t lb);
int c_dbcsr_acc_host_mem_allocate(void** host_mem, size_t nbytes, void* stream);
int c_dbcsr_acc_host_mem_deallocate(void* host_mem, void* stream);
int c_dbcsr_acc_memcpy_h2d(const void* host_mem, void* dev_mem, size_t nbytes, void* stream);
int c_dbcsr_acc_memcpy_d2h(const void* dev_mem, void* host_mem, size_t nbytes, void* stream);
int c_dbcsr_acc_memcpy_d2d(const void* devmem_src, void* devmem_dst, size_t nbytes, void* stream);
int c_dbcsr_acc_memset_zero(void* dev_mem, size_t offset, size_t nbytes, void* stream);
int c_dbcsr_acc_dev_mem_info(size_t* mem_free, size_t* mem_total);

/* acc.h:75-76 -- IMPORTED from the host (src/acc/dbcsr_acc_timings.F:23-51).
 * The library carries weak no-op definitions so that it also links without a
 * Fortran host; a host that defines them overrides the weak ones. */
void c_dbcsr_timeset(const char** routineN, const int* routineN_len, int* handle);
void c_dbcsr_timestop(const int* handle);

#if defined(__cplusplus)
}
#endif
#endif
