// ubench_fabric.hip -- what the memory system of an MI355X delivers to the block-product dataflow of
// BASELINE config 2: every block product pulls one A block (4232 B, from a window that stays
// in the XCD's L2) and one B block (4232 B, from a window that is too large for L2) into LDS and does nothing
// else.  The B window size selects where B comes from: <= 2 MB the 4 MB L2, <= 192 MB the 256 MB Infinity Cache
// (L2 <-> Infinity Cache fabric), larger HBM.  Reported: bytes delivered into LDS per second, by window and by
// staging method (LDS-DMA ring of S slots / register-staged loads + ds_write as the round-1 kernels did).
// Development tool, not part of the product library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_fabric ubench_fabric.hip && ./ubench_fabric
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../dbcsr_amd/csrc/dma_lds.h"

using namespace dbcsr_amd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int BLK = 4232, SLOTB = 4240;  // a 23 x 23 fp64 block; LDS slot rounded to 16 B
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// one wave per workgroup; `iters` products per wave; A block from window A (nblk_a blocks, private per XCD slice),
// B block from window B (nblk_b blocks); MODE 0: LDS-DMA ring of S slots, MODE 1: register-staged.
template <int S, int MODE, bool WITH_A>
__global__ void __launch_bounds__(64) stream_blocks(const char* __restrict__ a, unsigned nblk_a, const char* __restrict__ b,
                                                    unsigned nblk_b, int iters, int seq, double* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x, voff = lane * 16;
  const unsigned wave = blockIdx.x, xcd = blockIdx.x & 7;
  const unsigned lds0 = lds_offset_of(smem);
  auto blk_a = [&](int it) { return (size_t)((hash32(wave * 7919u + it) % (nblk_a / 8)) + xcd * (nblk_a / 8)) * BLK; };
  auto blk_b = [&](int it) {
    const unsigned r = seq ? (wave * (unsigned)iters + it) : hash32(wave * 104729u + it * 31u + 17u);
    return (size_t)(r % nblk_b) * BLK;
  };
  double acc = 0.0;
  if (MODE == 0) {
    auto issue = [&](int it, int slot) {
      const unsigned lds = lds0 + (unsigned)slot * 2 * SLOTB;
      if (WITH_A) dma_block<BLK>(a + blk_a(it), lds, voff);
      dma_block<BLK>(b + blk_b(it), lds + SLOTB, voff);
    };
    constexpr int PIECES = WITH_A ? 10 : 5;
    for (int j = 0; j < S - 1; ++j) issue(j, j);
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
      int sn = slot + S - 1; sn = sn >= S ? sn - S : sn;
      issue(it + S - 1, sn);  // (runs past iters: harmless extra loads, keeps the wait count constant)
      dma_wait<(S - 1) * PIECES>();
      acc += reinterpret_cast<const double*>(smem + slot * 2 * SLOTB + SLOTB)[lane];  // touch what landed
      slot = slot + 1 == S ? 0 : slot + 1;
    }
    dma_wait<0>();
  } else {
    u32x4 ra[5], rb[5];
    auto issue = [&](int it) {
      const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a + blk_a(it)), 0, BLK, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b + blk_b(it)), 0, BLK, 0x00020000);
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        if (WITH_A) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
        rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
      }
    };
    issue(0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        if (WITH_A) *reinterpret_cast<u32x4*>(smem + c * 1024 + voff) = ra[c];
        *reinterpret_cast<u32x4*>(smem + 5120 + c * 1024 + voff) = rb[c];
      }
      issue(it + 1);
      acc += reinterpret_cast<const double*>(smem + 5120)[lane];
    }
  }
  if (acc == 123.456) sink[0] = acc;
}

template <int S, int MODE, bool WITH_A>
static double run(const char* a, unsigned nblk_a, const char* b, unsigned nblk_b, int waves, int iters, int seq, size_t lds, double* sink) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((stream_blocks<S, MODE, WITH_A>), dim3(waves), dim3(64), lds, 0, a, nblk_a, b, nblk_b, iters, seq, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((stream_blocks<S, MODE, WITH_A>), dim3(waves), dim3(64), lds, 0, a, nblk_a, b, nblk_b, iters, seq, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  const size_t bufbytes = 3ull << 30;
  char *a, *b;
  double* sink;
  CK(hipMalloc(&a, 64ull << 20)); CK(hipMalloc(&b, bufbytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0, 64ull << 20)); CK(hipMemset(b, 0, bufbytes));
  const int iters = 14;  // products per wave, as config 2's C blocks
  const int waves = 2 * 1024 * 1024;  // ~ config 2's 2.03 M C blocks
  const double nprod = (double)waves * iters;
  printf("# %d waves x %d products, block %d B; A window 16 MB (2 MB per XCD slice), B window as listed\n", waves, iters, BLK);
  printf("# method               lds/wave  Bwin_MB  seq  ms      B_GB/s   total_into_LDS_GB/s\n");
  const unsigned nblk_a = (16u << 20) / BLK;
  const double winmb[] = {1, 4, 32, 128, 160, 512, 2048};
  for (double w : winmb) {
    const unsigned nb = (unsigned)(w * 1048576.0 / BLK);
    for (int seq = 0; seq < 2; ++seq) {
      if (seq && w < 2048) continue;
      struct R { const char* name; size_t lds; double ms; bool with_a; };
      std::vector<R> rs;
      rs.push_back({"dma S=2 A+B (9 w/CU)", 2 * 2 * SLOTB, run<2, 0, true>(a, nblk_a, b, nb, waves, iters, seq, 2 * 2 * SLOTB, sink), true});
      rs.push_back({"dma S=3 A+B (6 w/CU)", 3 * 2 * SLOTB, run<3, 0, true>(a, nblk_a, b, nb, waves, iters, seq, 3 * 2 * SLOTB, sink), true});
      rs.push_back({"dma S=4 A+B (4 w/CU)", 4 * 2 * SLOTB, run<4, 0, true>(a, nblk_a, b, nb, waves, iters, seq, 4 * 2 * SLOTB, sink), true});
      rs.push_back({"dma S=3 B only (6 w/CU)", 3 * 2 * SLOTB, run<3, 0, false>(a, nblk_a, b, nb, waves, iters, seq, 3 * 2 * SLOTB, sink), false});
      rs.push_back({"dma S=4 B only (4 w/CU)", 4 * 2 * SLOTB, run<4, 0, false>(a, nblk_a, b, nb, waves, iters, seq, 4 * 2 * SLOTB, sink), false});
      rs.push_back({"reg+ds_write A+B (16 w/CU)", 10240, run<1, 1, true>(a, nblk_a, b, nb, waves, iters, seq, 10240, sink), true});
      rs.push_back({"reg+ds_write A+B (8 w/CU)", 20480, run<1, 1, true>(a, nblk_a, b, nb, waves, iters, seq, 20480, sink), true});
      for (auto& r : rs)
        printf("%-28s %7zu  %7.0f  %d   %7.3f  %7.0f  %7.0f\n", r.name, r.lds, w, seq, r.ms, nprod * BLK / r.ms * 1e-6,
               nprod * BLK * (r.with_a ? 2 : 1) / r.ms * 1e-6);
    }
  }
  return 0;
}
