// ubench_c_write.hip -- what the C write pattern of the exact-size kernels costs by itself (BASELINE config 4: 14.3 M blocks of 23 x 23
// doubles, 60.7 GB, 1.3 products each).  One wave per block, the block leaves in 1 KiB pieces of 16 B per lane through a bounds-checked
// buffer store with the streaming hint, as in mm_exact.h / mm_dma.h.  Variants: block stride in doubles (529 = packed as DBCSR's data
// areas are, 530 = 16-byte aligned, 544 = 128-byte aligned) and bytes written per block (4232 = the block, 4352 = whole lines).
//   hipcc -O3 --offload-arch=gfx950 ubench_c_write.hip -o ubench_c_write && ./ubench_c_write [nblocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ void __launch_bounds__(256) write_blocks(double* c, long nblk, int stride, int bytes, int order_xcd) {
  long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (order_xcd) {  // consecutive blocks stay on one XCD (workgroup w runs on XCD w % 8): the engine's launch order
    const long wg = blockIdx.x, per = (gridDim.x + 7) / 8;
    b = ((wg & 7) * per + (wg >> 3)) * 4 + (threadIdx.x >> 6);
  }
  if (b >= nblk) return;
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(c + b * stride), 0, bytes, 0x00020000);
  const u32x4 v = {(unsigned)b, (unsigned)lane, 1u, 2u};
#pragma unroll
  for (int p = 0; p < 5; ++p) __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, p * 1024, AUX);
}

__global__ void __launch_bounds__(256) empty_waves(long nblk, int* sink) {
  if ((long)blockIdx.x * 4 + (threadIdx.x >> 6) == nblk + 7) *sink = 1;
}

int main(int argc, char** argv) {
  const long nblk = argc > 1 ? atol(argv[1]) : 14300000L;
  double* c = nullptr;
  if (hipMalloc(&c, (size_t)nblk * 544 * 8 + 8192) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  int* sink = nullptr;
  (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const unsigned grid = (unsigned)((nblk + 3) / 4);
  auto timeit = [&](const char* name, auto launch, double gb) {
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      (void)hipEventRecord(e0, 0);
      launch();
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms < best) best = ms;
    }
    printf("%-58s %8.3f ms  %7.2f TB/s\n", name, best, gb / best);
  };
  printf("# %ld blocks, one wave each\n", nblk);
  timeit("waves only (no store)", [&] { hipLaunchKernelGGL(empty_waves, dim3(grid), dim3(256), 0, 0, nblk, sink); }, 0.0);
  const int strides[] = {529, 530, 544, 544};
  const int bytes[] = {4232, 4232, 4232, 4352};
  for (int ord = 0; ord < 2; ++ord)
    for (int v = 0; v < 4; ++v) {
      char name[128];
      snprintf(name, sizeof name, "stride %d doubles, %d B per block, streaming, %s", strides[v], bytes[v], ord ? "XCD-contiguous" : "round-robin");
      const int s = strides[v], by = bytes[v];
      timeit(name, [&] { hipLaunchKernelGGL(write_blocks<2>, dim3(grid), dim3(256), 0, 0, c, nblk, s, by, ord); }, nblk * (double)by * 1e-9);
      snprintf(name, sizeof name, "stride %d doubles, %d B per block, plain,     %s", strides[v], bytes[v], ord ? "XCD-contiguous" : "round-robin");
      timeit(name, [&] { hipLaunchKernelGGL(write_blocks<0>, dim3(grid), dim3(256), 0, 0, c, nblk, s, by, ord); }, nblk * (double)by * 1e-9);
    }
  return 0;
}
