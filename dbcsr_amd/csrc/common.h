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

// dbcsr_amd_multiply's k-pass decision for the operand an engine saw last (mm_api.hip: k_passes).  Lives in the engine, so it dies with the
// handle; an operand whose index_stamp is 0 ("unknown generation", include/dbcsr_amd_mm.h) is never remembered.
struct KPassMemo {
  const void *row_p = nullptr, *blk_p = nullptr;
  int64_t nblks = -1, b_nblks = -1;   // (B's block count: its fill enters the estimate of the products per C block)
  uint64_t stamp = 0;
  int dt = 0, nblkrows = 0, nblkcols = 0, npass = 1;
};
KPassMemo* engine_kpass_memo(void* handle);  // mm_engine.hip

}  // namespace dbcsr_amd

#define ACC_CHECK(call)                                                        \
  do {                                                                         \
    hipError_t acc_e_ = (call);                                                \
    if (acc_e_ != hipSuccess) return dbcsr_amd::check(acc_e_, #call, __FILE__, __LINE__); \
  } while (0)

#endif
