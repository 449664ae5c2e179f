// fabric_probe.hip -- what the L2 <-> Infinity-Cache fabric of THIS device delivers, measured in well under a second: the ceiling the
// block-product kernels of BASELINE config 2 run against (DESIGN: one B block per product crosses that fabric).  Two readings, both from
// a window that is far too large for the 4 MB L2s and fits the 256 MB Infinity Cache:
//   stream : a plain read of the window, 16 bytes per lane, all CUs
//   gather : what a block product does to the memory system and nothing else -- every wave pulls 4232-byte blocks from pseudo-random
//            places of the window into LDS by LDS-DMA (ring of three slots, 16 waves per CU)
// bench.py puts the larger one next to the kernel's own fabric rate (roofline.fabric).  Measurement helper of the library, not on any
// product path.
#include "common.h"
#include "dma_lds.h"
#include "../../include/dbcsr_amd_mm.h"

namespace dbcsr_amd {

typedef unsigned int probe_u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) probe_stream_read(const probe_u32x4* __restrict__ p, size_t n16, int reps, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
      const probe_u32x4 v = p[i];
      acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
  if (acc == 0x12345u) sink[0] = acc;
}

__device__ __forceinline__ unsigned probe_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

constexpr int kProbeBlk = 4232, kProbeSlot = 4240, kProbeRing = 3;

__global__ void __launch_bounds__(64) probe_gather_blocks(const char* __restrict__ b, unsigned nblk, int iters, double* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) char smem[kProbeRing * kProbeSlot + 64];
  const int lane = threadIdx.x & 63, voff = lane * 16;
  const unsigned wave = blockIdx.x, lds0 = lds_offset_of(smem);
  auto issue = [&](int it, int slot) {
    const size_t at = (size_t)(probe_hash(wave * 104729u + it * 31u + 17u) % nblk) * kProbeBlk;
    dma_block<kProbeBlk>(b + at, lds0 + (unsigned)slot * kProbeSlot, voff);
  };
  constexpr int PIECES = (kProbeBlk + 1023) / 1024;
  for (int j = 0; j < kProbeRing - 1; ++j) issue(j, j);
  int slot = 0;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    int sn = slot + kProbeRing - 1;
    sn = sn >= kProbeRing ? sn - kProbeRing : sn;
    issue(it + kProbeRing - 1, sn);  // (runs past iters: harmless extra loads, keeps the wait count constant)
    dma_wait<(kProbeRing - 1) * PIECES>();
    acc += reinterpret_cast<const double*>(smem + slot * kProbeSlot)[lane];  // touch what landed
    slot = slot + 1 == kProbeRing ? 0 : slot + 1;
  }
  dma_wait<0>();
  if (acc == 123.456) sink[0] = acc;
}

}  // namespace dbcsr_amd

using namespace dbcsr_amd;

extern "C" int dbcsr_amd_fabric_probe(double* stream_tb_per_s, double* gather_tb_per_s) {
  const size_t window = (size_t)160 << 20;  // the production kernel's B panels are of this order (DBCSR_AMD_MM_PANEL_MB)
  char* buf = nullptr;
  unsigned* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 0;
  auto fail = [&](hipError_t e, const char* what) { rc = check(e, what, __FILE__, __LINE__); };
  hipError_t e;
  if ((e = hipMalloc(reinterpret_cast<void**>(&buf), window + 8192)) != hipSuccess) return check(e, "hipMalloc(probe window)", __FILE__, __LINE__);
  if ((e = hipMalloc(reinterpret_cast<void**>(&sink), 64)) != hipSuccess) fail(e, "hipMalloc(probe sink)");
  if (!rc && (e = hipMemset(buf, 0, window + 8192)) != hipSuccess) fail(e, "hipMemset(probe window)");
  if (!rc && ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess)) fail(e, "hipEventCreate");
  float ms = 0.f;
  if (!rc) {
    const int reps = 40;  // 6.4 GB: about a millisecond
    const size_t n16 = window / 16;
    hipLaunchKernelGGL(probe_stream_read, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const probe_u32x4*>(buf), n16, 2, sink);  // (fills the Infinity Cache)
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe_stream_read, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const probe_u32x4*>(buf), n16, reps, sink);
    (void)hipEventRecord(e1, 0);
    if ((e = hipEventSynchronize(e1)) != hipSuccess || (e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) fail(e, "probe_stream_read");
    if (!rc && stream_tb_per_s) *stream_tb_per_s = (double)window * reps / (ms * 1e-3) / 1e12;
  }
  if (!rc) {
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned nwaves = (unsigned)n_cu * 16u * 4u;  // four rounds of 16 one-wave workgroups per CU
    const int iters = 96;
    const unsigned nblk = (unsigned)(window / kProbeBlk);
    hipLaunchKernelGGL(probe_gather_blocks, dim3(nwaves), dim3(64), 0, 0, buf, nblk, 8, reinterpret_cast<double*>(sink));
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe_gather_blocks, dim3(nwaves), dim3(64), 0, 0, buf, nblk, iters, reinterpret_cast<double*>(sink));
    (void)hipEventRecord(e1, 0);
    if ((e = hipEventSynchronize(e1)) != hipSuccess || (e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) fail(e, "probe_gather_blocks");
    // whole 128-byte lines cross the fabric: a 4232-byte block that starts anywhere touches 34-35 of them (4416 bytes on average)
    if (!rc && gather_tb_per_s) *gather_tb_per_s = (double)nwaves * (iters + kProbeRing - 1) * 4416.0 / (ms * 1e-3) / 1e12;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (sink) (void)hipFree(sink);
  (void)hipFree(buf);
  return rc;
}
