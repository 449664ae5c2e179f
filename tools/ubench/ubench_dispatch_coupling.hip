// ubench_dispatch_coupling.hip -- groundwork for a mixed launch: workgroup b of a grid runs on XCD b % 8.  If two XCDs are
// almost full with a persistent kernel (one 512-thread workgroup per CU, 136 KB of LDS, 208 registers per lane: what the tile kernel
// takes) and the workgroups a second grid sends there only have to START and leave, does their slower turnover hold back the
// workgroups of the same grid on the six free XCDs (in-order round-robin dispatch), or not?
//   grid under test: N workgroups of 64 threads, 9.5 KB of LDS, 88 registers (the exact-size kernel's footprint); on the XCDs of
//   `off_mask` they leave at once, elsewhere they stay ~`work_us` microseconds.
//   hipcc -O3 --offload-arch=gfx950 ubench_dispatch_coupling.hip -o ubench_dispatch_coupling && ./ubench_dispatch_coupling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(512) occupier(volatile int* stop, unsigned xcd_mask) {
  extern __shared__ char lds[];
  asm volatile("v_mov_b32 v207, 0" ::: "v207");   // 208 registers per lane
  if (!((xcd_mask >> (blockIdx.x & 7)) & 1u)) return;
  lds[threadIdx.x] = 1;
  while (!*stop) __builtin_amdgcn_s_sleep(64);
}

__global__ void __launch_bounds__(64) small_work(unsigned off_mask, int work_ticks, unsigned* sink) {
  extern __shared__ char lds[];
  asm volatile("v_mov_b32 v87, 0" ::: "v87");     // 88 registers per lane
  if ((off_mask >> (blockIdx.x & 7)) & 1u) return;
  lds[threadIdx.x] = 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)work_ticks) __builtin_amdgcn_s_sleep(8);
  if (lds[threadIdx.x] == 7) sink[0] = 1;
}

int main(int argc, char** argv) {
  const unsigned n = argc > 1 ? (unsigned)atol(argv[1]) : 2000000u;   // workgroups of the grid under test (config 2: 2.03 M C blocks)
  const int work_ticks = 900;                                          // 9 us per workgroup on the XCDs that work
  int* stop;
  (void)hipHostMalloc(reinterpret_cast<void**>(&stop), sizeof(int), hipHostMallocMapped);
  unsigned* sink;
  (void)hipMalloc(&sink, 4);
  hipStream_t s1, s2;
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(occupier), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("# %u workgroups of 64 threads (9.5 KB LDS, 88 VGPRs), 9 us each on the XCDs that work\n", n);
  printf("# XCDs where the grid's workgroups leave at once | persistent occupier on those XCDs | ms\n");
  for (unsigned off : {0x00u, 0x11u, 0x01u}) {
    for (int occ = 0; occ < 2; ++occ) {
      if (occ && !off) continue;
      *stop = 0;
      if (occ) {
        hipLaunchKernelGGL(occupier, dim3(256), dim3(512), 136 * 1024, s1, stop, off);
        // give it time to settle on its CUs
        for (volatile int w = 0; w < 20000000; ++w) {}
      }
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, s2);
        hipLaunchKernelGGL(small_work, dim3(n), dim3(64), 9728, s2, off, work_ticks, sink);
        (void)hipEventRecord(e1, s2);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      *stop = 1;
      (void)hipStreamSynchronize(s1);
      printf("0x%02x  %s  %8.3f\n", off, occ ? "yes" : "no ", best);
    }
  }
  return 0;
}
