// ubench_fabric.hip -- what the memory system of an MI355X delivers to the block-product dataflow of
// BASELINE config 2: every block product pulls one A block (4232 B, from a window that stays
// in the XCD's L2) and one B block (4232 B, from a window that is too large for L2) into LDS and does nothing
// else.  The B window size selects where B comes from: <= 2 MB the 4 MB L2, <= 192 MB the 256 MB Infinity Cache
// (L2 <-> Infinity Cache fabric), larger HBM.  Reported: bytes delivered into LDS per second, by window and by
// staging method (LDS-DMA ring of S slots / register-staged loads + ds_write as the round-1 kernels did).
// Further sections: cache-policy bits on the B stream, 128-byte aligned blocks, how the rate scales with the number
// of CUs that take part (per-CU limit or shared limit?), a plain streaming read for reference, and the C-block
// write pattern of BASELINE config 4 (4232-byte blocks, packed or padded to whole cache lines).
// Development tool, not part of the product library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_fabric ubench_fabric.hip && ./ubench_fabric [section...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// NW waves per workgroup, each with its own ring; `iters` products per wave; A block from window A (nblk_a blocks, one
// slice per XCD), B block from window B (nblk_b blocks of `stride` bytes); MODE 0: LDS-DMA ring of S slots, 1: register-staged.
template <int S, int MODE, bool WITH_A, int POLB, int NW>
__global__ void __launch_bounds__(64 * NW) stream_blocks(const char* __restrict__ a, unsigned nblk_a, const char* __restrict__ b,
                                                         unsigned nblk_b, int stride, int iters, int seq, double* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int lane = threadIdx.x & 63, voff = lane * 16;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr int RING = MODE == 0 ? S * 2 * SLOTB : 10240;
  char* smem = smem_all + wid * RING;
  const unsigned wave = blockIdx.x * NW + wid, xcd = blockIdx.x & 7;
  const unsigned lds0 = lds_offset_of(smem);
  auto blk_a = [&](int it) { return (size_t)((hash32(wave * 7919u + it) % (nblk_a / 8)) + xcd * (nblk_a / 8)) * stride; };
  auto blk_b = [&](int it) {
    const unsigned r = seq ? (wave * (unsigned)iters + it) : hash32(wave * 104729u + it * 31u + 17u);
    return (size_t)(r % nblk_b) * stride;
  };
  double acc = 0.0;
  if (MODE == 0) {
    auto issue = [&](int it, int slot) {
      const unsigned lds = lds0 + (unsigned)slot * 2 * SLOTB;
      if (WITH_A) dma_block<BLK>(a + blk_a(it), lds, voff);
      dma_block<BLK, POLB>(b + blk_b(it), lds + SLOTB, voff);
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

// plain streaming read of `bytes` bytes, 16 B per lane, grid-stride, `reps` passes
__global__ void __launch_bounds__(256) stream_read(const u32x4* __restrict__ p, size_t n16, int reps, double* __restrict__ sink) {
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
      const u32x4 v = p[i];
      acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
  if (acc == 0x12345u) sink[0] = acc;
}

// the same read by the workgroups of SOME XCDs only (workgroup b runs on XCD b % 8; the others leave at once): is the L2 <-> fabric
// ceiling a sum of per-XCD limits or one shared limit?
__global__ void __launch_bounds__(256) stream_read_xcds(const u32x4* __restrict__ p, size_t n16, int reps, unsigned xcd_mask, int nactive,
                                                        double* __restrict__ sink) {
  const unsigned xcd = blockIdx.x & 7u;
  if (!((xcd_mask >> xcd) & 1u)) return;
  const unsigned rank = __popc(xcd_mask & ((1u << xcd) - 1u));          // position of this XCD among the active ones
  const size_t wg = (size_t)(blockIdx.x >> 3) * nactive + rank, nwg = (size_t)(gridDim.x >> 3) * nactive;
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r)
    for (size_t i = wg * 256 + threadIdx.x; i < n16; i += nwg * 256) {
      const u32x4 v = p[i];
      acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
  if (acc == 0x12345u) sink[0] = acc;
}

// C-block write pattern: wave w writes block w (4232 B payload) at w * stride, in 1 KiB pieces of 16 B per lane
template <int AUX>
__global__ void __launch_bounds__(256) write_blocks(char* __restrict__ c, size_t nblk, int stride) {
  const int lane = threadIdx.x & 63, voff = lane * 16;
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= nblk) return;
  const unsigned long long base = (unsigned long long)(c + w * (size_t)stride);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)base)), 0, BLK,
      0x00020000);
  const u32x4 v = {(unsigned)w, (unsigned)lane, 1u, 2u};
#pragma unroll
  for (int cc = 0; cc < 5; ++cc) __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, cc * 1024, AUX);
}

static float timed(void (*launch)(void*), void* ctx) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(ctx);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  launch(ctx);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms;
}

struct Ctx { const char* a; unsigned nblk_a; const char* b; unsigned nblk_b; int stride; int waves; int iters; int seq; size_t lds; double* sink; };

template <int S, int MODE, bool WITH_A, int POLB, int NW>
static void launch_sb(void* p) {
  Ctx* c = (Ctx*)p;
  hipLaunchKernelGGL((stream_blocks<S, MODE, WITH_A, POLB, NW>), dim3(c->waves / NW), dim3(64 * NW), c->lds, 0, c->a, c->nblk_a, c->b, c->nblk_b,
                     c->stride, c->iters, c->seq, c->sink);
}

static bool want(int argc, char** argv, const char* name) {
  if (argc <= 1) return true;
  for (int i = 1; i < argc; ++i)
    if (!strcmp(argv[i], name)) return true;
  return false;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t bufbytes = 3ull << 30;
  char *a, *b;
  double* sink;
  CK(hipMalloc(&a, 64ull << 20)); CK(hipMalloc(&b, bufbytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0, 64ull << 20)); CK(hipMemset(b, 0, bufbytes));
  const int iters = 14;               // products per wave, as config 2's C blocks
  const int waves = 2 * 1024 * 1024;  // ~ config 2's 2.03 M C blocks
  const double nprod = (double)waves * iters;
  const unsigned nblk_a = (16u << 20) / 4352;
  auto row = [&](const char* name, double w, int seq, float ms, bool with_a, double np) {
    printf("%-44s %7.0f  %d   %7.3f  %7.0f  %7.0f\n", name, w, seq, ms, np * BLK / ms * 1e-6, np * BLK * (with_a ? 2 : 1) / ms * 1e-6);
  };
  if (want(argc, argv, "windows")) {
    printf("# [windows] %d waves x %d products, block %d B packed; A window 16 MB (2 MB per XCD slice), B window as listed\n", waves, iters, BLK);
    printf("# method                                      Bwin_MB  seq  ms      B_GB/s   total_into_LDS_GB/s\n");
    const double winmb[] = {1, 4, 32, 160, 512, 2048};
    for (double w : winmb) {
      Ctx c{a, nblk_a, b, (unsigned)(w * 1048576.0 / BLK), BLK, waves, iters, 0, 0, sink};
      c.lds = 2 * 2 * SLOTB; row("dma S=2 A+B (9 w/CU)", w, 0, timed(launch_sb<2, 0, true, 0, 1>, &c), true, nprod);
      c.lds = 3 * 2 * SLOTB; row("dma S=3 A+B (6 w/CU)", w, 0, timed(launch_sb<3, 0, true, 0, 1>, &c), true, nprod);
      c.lds = 3 * 2 * SLOTB; row("dma S=3 B only (6 w/CU)", w, 0, timed(launch_sb<3, 0, false, 0, 1>, &c), false, nprod);
      c.lds = 10240; row("reg+ds_write A+B (16 w/CU)", w, 0, timed(launch_sb<1, 1, true, 0, 1>, &c), true, nprod);
      c.lds = 20480; row("reg+ds_write A+B (8 w/CU)", w, 0, timed(launch_sb<1, 1, true, 0, 1>, &c), true, nprod);
      c.lds = 10240; row("reg+ds_write B only (16 w/CU)", w, 0, timed(launch_sb<1, 1, false, 0, 1>, &c), false, nprod);
    }
  }
  if (want(argc, argv, "policy")) {
    printf("# [policy] cache-policy bits on the B stream (B window 160 MB, dma S=2, 9 w/CU), and 128-byte aligned blocks (stride 4352)\n");
    Ctx c{a, nblk_a, b, (unsigned)(160.0 * 1048576.0 / BLK), BLK, waves, iters, 0, 2 * 2 * SLOTB, sink};
    row("A+B, B default", 160, 0, timed(launch_sb<2, 0, true, 0, 1>, &c), true, nprod);
    row("A+B, B nt", 160, 0, timed(launch_sb<2, 0, true, 1, 1>, &c), true, nprod);
    row("A+B, B sc1", 160, 0, timed(launch_sb<2, 0, true, 2, 1>, &c), true, nprod);
    row("A+B, B sc0 sc1", 160, 0, timed(launch_sb<2, 0, true, 3, 1>, &c), true, nprod);
    Ctx d = c; d.stride = 4352; d.nblk_b = (unsigned)(160.0 * 1048576.0 / 4352);
    row("A+B, blocks 128-B aligned (stride 4352)", 160, 0, timed(launch_sb<2, 0, true, 0, 1>, &d), true, nprod);
    d.lds = 10240; row("reg+ds_write A+B aligned (16 w/CU)", 160, 0, timed(launch_sb<1, 1, true, 0, 1>, &d), true, nprod);
    d.lds = 3 * 2 * SLOTB; row("B only aligned (dma S=3)", 160, 0, timed(launch_sb<3, 0, false, 0, 1>, &d), false, nprod);
  }
  if (want(argc, argv, "cus")) {
    printf("# [cus] one 6-wave workgroup per CU (LDS 149 KB), G workgroups, each wave %d products: does the rate follow the number of CUs?\n", 14 * 64);
    for (int G : {32, 64, 128, 256, 512}) {
      for (int with_a = 0; with_a < 2; ++with_a) {
        Ctx c{a, nblk_a, b, (unsigned)(160.0 * 1048576.0 / BLK), BLK, G * 6, 14 * 64, 0, (size_t)6 * 3 * 2 * SLOTB, sink};
        char nm[64];
        snprintf(nm, sizeof nm, "G=%d workgroups x 6 waves, %s", G, with_a ? "A+B" : "B only");
        const double np = (double)G * 6 * 14 * 64;
        row(nm, 160, 0, with_a ? timed(launch_sb<3, 0, true, 0, 6>, &c) : timed(launch_sb<3, 0, false, 0, 6>, &c), with_a, np);
      }
    }
  }
  if (want(argc, argv, "stream")) {
    printf("# [stream] plain streaming read, 16 B per lane, 2048 workgroups x 256, passes over the window\n# window_MB  ms  GB/s\n");
    for (double w : {2.0, 16.0, 64.0, 128.0, 192.0, 1024.0, 3072.0}) {
      const size_t n16 = (size_t)(w * 1048576.0) / 16;
      const int reps = w < 1024 ? (int)(4096 / w) : 2;
      struct SC { const u32x4* p; size_t n; int reps; double* sink; } sc{(const u32x4*)b, n16, reps, sink};
      auto l = [](void* q) { SC* s = (SC*)q; hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, s->p, s->n, s->reps, s->sink); };
      const float ms = timed(l, &sc);
      printf("%8.0f  %8.3f  %8.0f\n", w, ms, (double)n16 * 16 * reps / ms * 1e-6);
    }
  }
  if (want(argc, argv, "xcds")) {
    printf("# [xcds] plain streaming read of a 1 GB window (Infinity Cache / HBM) and of 128 MB (Infinity Cache), 16 B per lane, by the workgroups of some XCDs only\n");
    printf("# window_MB  active XCDs (mask)  ms  GB/s  GB/s per active XCD\n");
    for (double w : {128.0, 1024.0})
      for (unsigned mask : {0xffu, 0x0fu, 0x55u, 0x03u, 0x01u}) {
        const size_t n16 = (size_t)(w * 1048576.0) / 16;
        const int reps = w < 1024 ? 16 : 2;
        const int na = __builtin_popcount(mask);
        struct SC { const u32x4* p; size_t n; int reps; unsigned mask; int na; double* sink; } sc{(const u32x4*)b, n16, reps, mask, na, sink};
        auto l = [](void* q) { SC* s = (SC*)q; hipLaunchKernelGGL(stream_read_xcds, dim3(2048), dim3(256), 0, 0, s->p, s->n, s->reps, s->mask, s->na, s->sink); };
        const float ms = timed(l, &sc);
        const double gbs = (double)n16 * 16 * reps / ms * 1e-6;
        printf("%8.0f  %d (0x%02x)  %8.3f  %8.0f  %8.0f\n", w, na, mask, ms, gbs, gbs / na);
      }
  }
  if (want(argc, argv, "write")) {
    printf("# [write] C-block write pattern of config 4: 14.3 M blocks of 4232 B, wave w -> block w\n# layout  aux  ms  GB/s(payload)\n");
    const size_t nblk = 14300000;
    char* cbuf;
    CK(hipMalloc(&cbuf, nblk * 4352 + 4096));
    struct WC { char* c; size_t n; int stride; } wc{cbuf, nblk, BLK};
    for (int stride : {BLK, 4352}) {
      wc.stride = stride;
      auto l0 = [](void* q) { WC* s = (WC*)q; hipLaunchKernelGGL(write_blocks<0>, dim3((unsigned)((s->n + 3) / 4)), dim3(256), 0, 0, s->c, s->n, s->stride); };
      auto l2 = [](void* q) { WC* s = (WC*)q; hipLaunchKernelGGL(write_blocks<2>, dim3((unsigned)((s->n + 3) / 4)), dim3(256), 0, 0, s->c, s->n, s->stride); };
      float ms = timed(l0, &wc);
      printf("stride %d  plain  %8.3f  %8.0f\n", stride, ms, (double)nblk * BLK / ms * 1e-6);
      ms = timed(l2, &wc);
      printf("stride %d  nt     %8.3f  %8.0f\n", stride, ms, (double)nblk * BLK / ms * 1e-6);
    }
  }
  return 0;
}
