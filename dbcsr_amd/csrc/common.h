// common.h -- shared helpers of libdbcsr_acc_amd (gfx950 only).
#ifndef DBCSR_AMD_COMMON_H
#define DBCSR_AMD_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dbcsr_amd {

int check(hipError_t e, const char* what, const char* file, int line);

// A C-ABI stream handle is a pointer to a heap-allocated hipStream_t (see
// include/dbcsr_acc.h); NULL means the null stream.
static inline hipStream_t stream_of(void* handle) { return handle ? *static_cast<hipStream_t*>(handle) : (hipStream_t)0; }

// device memory through the caching allocator of the acc runtime (acc_runtime.hip): what c_dbcsr_acc_dev_mem_allocate / _deallocate
// use, for the matrices the library itself hands out (dbcsr_amd_multiply's result and temporaries)
hipError_t pool_malloc(void** p, size_t nbytes);
hipError_t pool_free(void* p);

}  // namespace dbcsr_amd

#define ACC_CHECK(call)                                                        \
  do {                                                                         \
    hipError_t acc_e_ = (call);                                                \
    if (acc_e_ != hipSuccess) return dbcsr_amd::check(acc_e_, #call, __FILE__, __LINE__); \
  } while (0)

#endif
