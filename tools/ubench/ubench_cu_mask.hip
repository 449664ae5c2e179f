// ubench_cu_mask.hip -- where do the workgroups of a kernel land when its stream was made with hipExtStreamCreateWithCUMask?
// (Groundwork for a mixed launch: two kernels that must stay on disjoint sets of XCDs)  For several masks: histogram of
// XCC_ID over a grid of 4096 small workgroups, and the relation between blockIdx and XCC_ID.
//   hipcc -O3 --offload-arch=gfx950 ubench_cu_mask.hip -o ubench_cu_mask && ./ubench_cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(64) where(unsigned* xcc, unsigned* hw, int spin) {
  unsigned x, h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  if (threadIdx.x == 0) xcc[blockIdx.x] = x & 0xf, hw[blockIdx.x] = h;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);  // keep the slot for a moment: the grid has to spread
}

int main() {
  int ncu = 0;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  printf("# %d CUs\n", ncu);
  const int n = 4096;
  unsigned *xcc, *hw;
  (void)hipMalloc(&xcc, n * 4);
  (void)hipMalloc(&hw, n * 4);
  struct M { const char* name; std::vector<uint32_t> bits; };
  std::vector<M> masks;
  const int words = (ncu + 31) / 32;
  auto make = [&](const char* name, auto pred) {
    M m{name, std::vector<uint32_t>(words, 0u)};
    for (int c = 0; c < ncu; ++c)
      if (pred(c)) m.bits[c / 32] |= 1u << (c % 32);
    masks.push_back(m);
  };
  make("all CUs", [](int) { return true; });
  make("CUs 0..31 (first 32 bits)", [](int c) { return c < 32; });
  make("CUs 0..127 (first half of the bits)", [&](int c) { return c < ncu / 2; });
  make("CU c with c % 8 == 0", [](int c) { return c % 8 == 0; });
  make("CU c with c % 8 < 2", [](int c) { return c % 8 < 2; });
  make("CU c with (c / 8) % 4 == 0", [](int c) { return (c / 8) % 4 == 0; });
  make("CU c with c % 2 == 0", [](int c) { return c % 2 == 0; });
  for (const M& m : masks) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)m.bits.size(), m.bits.data());
    if (e != hipSuccess) { printf("%-40s hipExtStreamCreateWithCUMask: %s\n", m.name, hipGetErrorString(e)); continue; }
    (void)hipMemsetAsync(xcc, 0xff, n * 4, st);
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, st, xcc, hw, 200);
    (void)hipStreamSynchronize(st);
    std::vector<unsigned> hx(n), hh(n);
    (void)hipMemcpy(hx.data(), xcc, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hh.data(), hw, n * 4, hipMemcpyDeviceToHost);
    int hist[16] = {0}, same = 0;
    for (int b = 0; b < n; ++b) {
      if (hx[b] < 16) hist[hx[b]]++;
      same += hx[b] == (unsigned)(b % 8);
    }
    printf("%-40s workgroups per XCC_ID:", m.name);
    for (int x = 0; x < 8; ++x) printf(" %5d", hist[x]);
    printf("   blockIdx %% 8 == XCC_ID for %d of %d\n", same, n);
    (void)hipStreamDestroy(st);
  }
  return 0;
}
