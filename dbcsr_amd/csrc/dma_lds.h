// dma_lds.h -- LDS-DMA primitives for gfx950 (buffer_load_dwordx4 ... lds), shared by the product kernels and the
// micro-benchmarks.  See mm_dma.h for the rules the inline asm follows.
#ifndef DBCSR_AMD_DMA_LDS_H
#define DBCSR_AMD_DMA_LDS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dbcsr_amd {

typedef unsigned int dma_rsrc_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ dma_rsrc_t dma_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  dma_rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);  // stride 0, no swizzle
  r[2] = bytes;                                                           // num_records: raw buffer, bytes
  r[3] = 0x00020000u;
  return r;
}

// one whole 1 KiB piece: lane l copies bytes [soff + 16 l, soff + 16 l + 16) of the buffer to LDS byte lds + 16 l
// (the first piece of a block opens with s_nop 4: its descriptor may come straight from v_readlane / v_readfirstlane,
// and a VALU-written SGPR needs 5 wait states before a VMEM instruction reads it -- the compiler does not pad inside asm)
// POL: cache policy bits of the load (0 default, 1 nt, 2 sc1, 3 sc0 sc1) -- used by the micro-benchmarks only
#define DBCSR_DMA_ASM(NOPS, MODS)                                                                                              \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop " NOPS "\n\tbuffer_load_dwordx4 %1, %2, %3 offen " MODS " lds" ::"s"(lds), "v"(voff), "s"(rsrc), \
               "s"(soff)                                                                                                       \
               : "memory")
template <bool FRESH, int POL = 0>
__device__ __forceinline__ void dma_piece(const dma_rsrc_t& rsrc, unsigned lds, int voff, unsigned soff) {
  if (FRESH) {
    if (POL == 0) DBCSR_DMA_ASM("4", "");
    if (POL == 1) DBCSR_DMA_ASM("4", "nt");
    if (POL == 2) DBCSR_DMA_ASM("4", "sc1");
    if (POL == 3) DBCSR_DMA_ASM("4", "sc0 sc1");
  } else {
    if (POL == 0) DBCSR_DMA_ASM("0", "");
    if (POL == 1) DBCSR_DMA_ASM("0", "nt");
    if (POL == 2) DBCSR_DMA_ASM("0", "sc1");
    if (POL == 3) DBCSR_DMA_ASM("0", "sc0 sc1");
  }
}
#undef DBCSR_DMA_ASM
// the same for the first `lanes` lanes only (last piece of a block): nothing is written past the block's end in LDS
#define DBCSR_DMA_ASM_M(MODS)                                                                                                  \
  asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 exec, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen " MODS                 \
               " lds\n\ts_mov_b64 exec, -1" ::"s"(lds),                                                                        \
               "v"(voff), "s"(rsrc), "s"(soff), "s"(mask)                                                                      \
               : "memory")
template <bool FRESH, int POL = 0>
__device__ __forceinline__ void dma_piece_masked(const dma_rsrc_t& rsrc, unsigned lds, int voff, unsigned soff, unsigned long long mask) {
  if (POL == 0) DBCSR_DMA_ASM_M("");
  if (POL == 1) DBCSR_DMA_ASM_M("nt");
  if (POL == 2) DBCSR_DMA_ASM_M("sc1");
  if (POL == 3) DBCSR_DMA_ASM_M("sc0 sc1");
}
#undef DBCSR_DMA_ASM_M

template <int BYTES, int C, int POL = 0>
__device__ __forceinline__ void dma_block_pieces(const dma_rsrc_t& rsrc, unsigned lds, int voff) {
  constexpr int NP = (BYTES + 1023) / 1024, REM = BYTES - 1024 * (NP - 1), LAST = (REM + 15) / 16;
  if constexpr (C < NP) {
    if constexpr (C < NP - 1 || LAST == 64)
      dma_piece<C == 0, POL>(rsrc, lds + 1024u * C, voff, 1024u * C);
    else
      dma_piece_masked<C == 0, POL>(rsrc, lds + 1024u * C, voff, 1024u * C, (1ull << (LAST & 63)) - 1ull);
    dma_block_pieces<BYTES, C + 1, POL>(rsrc, lds, voff);
  }
}

// a block of BYTES bytes at src -> LDS byte offset lds (as stored), in ceil(BYTES / 1024) DMA instructions
template <int BYTES, int POL = 0>
__device__ __forceinline__ void dma_block(const void* src, unsigned lds, int voff) {
  const dma_rsrc_t rsrc = dma_make_rsrc(src, (unsigned)BYTES);
  dma_block_pieces<BYTES, 0, POL>(rsrc, lds, voff);
}

template <int N_>
__device__ __forceinline__ void dma_wait() {
  static_assert(N_ >= 0 && N_ < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

__device__ __forceinline__ unsigned lds_offset_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

}  // namespace dbcsr_amd
#endif
