// mm_jit.h -- run-time compilation (hiprtc) of the exact-size class kernels of mm_exact.h and the stack kernels of smm_exact.h, cached per process.
// The reference does the same per (m, n, k) triple (src/acc/libsmm_acc/libsmm_acc.cpp:90-195, ~0.5 s per kernel there).
#ifndef DBCSR_AMD_MM_JIT_H
#define DBCSR_AMD_MM_JIT_H
#include <hip/hip_runtime.h>

namespace dbcsr_amd {

struct ClassKernel {
  hipFunction_t fn = nullptr;
  int wave_lds = 0;  // LDS bytes per wave (4 waves per workgroup)
};

// 0 on success; the kernel for C blocks of m x n whose products have inner sizes k0, k1, k2 (0 = absent, k0 > 0).
// Thread-safe; compiles on first use (~1 s), then served from the cache.  Non-zero when hiprtc is unavailable or fails
// (the caller then runs the generic kernel on that class).
// g > 1: the variant in which a wave walks g consecutive C blocks of the class (mm_class_stream_body)
int jit_class_kernel(int m, int n, int k0, int k1, int k2, int g, ClassKernel* out);


// The exact-size kernel of smm_exact.h for homogeneous parameter stacks of (m, n, k), all <= 32; bt: B as libsmm_acc_transpose leaves it
// (n x k).  Same contract as above: thread-safe, compiled on the first stack of a triplet, non-zero when hiprtc is unavailable or fails
// (the caller then runs smm_stack_f64_lds).
struct StackKernel {
  hipFunction_t fn = nullptr;
  int wave_lds = 0;
};
// compile = false: only a kernel another call (any thread) compiled before; 1 when there is none
int jit_stack_kernel(int m, int n, int k, bool bt, StackKernel* out, bool compile = true);

}  // namespace dbcsr_amd
#endif
