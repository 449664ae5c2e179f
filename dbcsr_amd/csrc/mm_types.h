// mm_types.h -- product-list records and launch helpers shared by the ahead-of-time kernels (mm_engine.hip), the LDS-DMA
// kernels (mm_dma.h) and the kernels compiled at run time for the (m, n, k) classes of a multiply (mm_exact.h, mm_jit.hip).
// Self-contained on purpose: this text is also handed to hiprtc.
#ifndef DBCSR_AMD_MM_TYPES_H
#define DBCSR_AMD_MM_TYPES_H
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else  // hiprtc: no system headers; the fixed-width types live in a private namespace there
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#endif

namespace dbcsr_amd {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Entry {  // one block product feeding a C block: 12 bytes, 40-bit element offsets (any operand that fits 288 GB)
  uint32_t a_lo, b_lo;  // low 32 bits of the element offsets into the A / B data areas
  uint32_t w;           // bits 0-15: k extent of this product; bits 16-23 / 24-31: bits 32-39 of the A / B offset
  __device__ __forceinline__ uint64_t a_off() const { return (uint64_t)a_lo | ((uint64_t)((w >> 16) & 0xffu) << 32); }
  __device__ __forceinline__ uint64_t b_off() const { return (uint64_t)b_lo | ((uint64_t)(w >> 24) << 32); }
  __device__ __forceinline__ int ks() const { return (int)(w & 0xffffu); }
  __device__ __forceinline__ static Entry make(int64_t a, int64_t b, int k) {
    Entry e;
    e.a_lo = (uint32_t)a;
    e.b_lo = (uint32_t)b;
    e.w = ((uint32_t)k & 0xffffu) | ((uint32_t)(((uint64_t)a >> 32) & 0xffu) << 16) | ((uint32_t)(((uint64_t)b >> 32) & 0xffu) << 24);
    return e;
  }
};

struct Desc {  // one C block
  int64_t c_off;       // element offset in C_out data
  int64_t cin_off;     // element offset in C_in data, -1 if the block is new
  int64_t prod_start;  // first Entry
  int32_t prod_cnt;
  int16_t m, n;
};

// The A image of a product may end inside the B image's first piece in LDS: the B pieces are stored AFTER the A pieces and overwrite that
// tail.  Every lane's own addresses are disjoint, so only the order of the two groups of ds_write instructions makes this correct
// (one wave, in-order LDS queue): the compiler must not move a B store above an A store.
#define DBCSR_AMD_LDS_ORDER() asm volatile("" ::: "memory")

struct Work {  // one position of the launch order: the C block's descriptor AND its first product, 48 bytes.  A wave reads
               // work[pos] (neighbouring waves read neighbouring records) and can request its first operands at once:
               // the dependent chain order[pos] -> descs[cb] -> entries[prod_start] -> operands becomes work[pos] -> operands
  int64_t c_off, cin_off, prod_start;
  int32_t prod_cnt;  // -1: padding position (no C block)
  int16_t m, n;
  uint32_t a_lo, b_lo, w;  // the first Entry of the block (undefined when prod_cnt == 0)
  int32_t cb;              // index of the C block (order[pos])
};

// the image of B in LDS of the fp32 direct kernel (mm_numeric_f32.h) and the lab's fp32 group kernel (mm_group.hip): B as stored (k contiguous),
// 32 rows of K + 4 floats
constexpr int F32D_ROWS = 32;
static inline constexpr int f32d_pitch(int K) { return K + 4; }
static inline constexpr int f32d_wave_floats(int K) { return F32D_ROWS * f32d_pitch(K); }

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // Workgroup b is dispatched to XCD b % 8 (MI355X_MICROARCH.md); give each XCD a
  // contiguous range of C blocks so that the A block-row it works on stays in
  // that XCD's L2.  Bijective for any nwg.  Speed only, never correctness.
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace dbcsr_amd
#endif
