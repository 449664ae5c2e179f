// Microbenchmarks + operand-layout probes for the fp64/fp32 matrix and vector
// pipes of gfx950.  Development tool (not part of the product library).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_fp64 ubench_fp64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE> __global__ void __launch_bounds__(256) rate_kernel(double* out, long long* cyc, int iters, double seed) {
  const int lane = threadIdx.x;
  double a = seed + lane * 1e-3, b = seed * 0.5 + lane * 1e-4;
  long long t0 = 0, t1 = 0;
  if (MODE == 0) {  // mfma f64 16x16x4, 4 independent accumulators
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (MODE == 1) {  // mfma f64 16x16x4, dependent chain
    d4 c0 = {0, 0, 0, 0};
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c0[1];
  } else if (MODE == 2) {  // mfma f64 4x4x4 (4 blocks), 4 independent
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
  } else if (MODE == 3) {  // mfma f64 4x4x4 dependent
    double c0 = 0;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0;
  } else if (MODE == 4) {  // v_fma_f64, 8 independent chains (4 "instructions" = 8 fma per iter -> count 8)
    double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_fma(a, b, c0); c1 = __builtin_fma(a, b, c1); c2 = __builtin_fma(a, b, c2); c3 = __builtin_fma(a, b, c3);
      c4 = __builtin_fma(a, b, c4); c5 = __builtin_fma(a, b, c5); c6 = __builtin_fma(a, b, c6); c7 = __builtin_fma(a, b, c7);
      asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  } else if (MODE == 5) {  // mfma f32 32x32x2, 2 independent
    float af = (float)a, bf = (float)b;
    f16v c0 = {0}, c1 = {0};
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c1, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1];
  } else if (MODE == 6) {  // mfma f32 16x16x4, 4 independent
    float af = (float)a, bf = (float)b;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c3, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  }
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// Layout probes: one-hot A lane la (value 1.0), B lane value = lane+1.
// T[la][lo*R + r] = D register r of output lane lo.
__global__ void probe_16x16x4(double* T) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    double a = (lane == la) ? 1.0 : 0.0, b = lane + 1.0;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) T[(la * 64 + lane) * 4 + r] = c[r];
  }
}
__global__ void probe_4x4x4(double* T) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    double a = (lane == la) ? 1.0 : 0.0, b = lane + 1.0;
    double c = 0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    T[la * 64 + lane] = c;
  }
}

template <int MODE> void run_rate(const char* name, double flop_per_inst, int inst_per_iter) {
  const int iters = 2000;
  double* out; long long* cyc;
  for (int wpb = 4; wpb <= 8; wpb += 4) {  // waves per block (per CU when grid=1): 1 or 2 per SIMD
    int threads = wpb * 64; (void)threads;
  }
  CK(hipMalloc(&out, sizeof(double) * 1024 * 2048));
  CK(hipMalloc(&cyc, sizeof(long long) * 16 * 2048));
  // (a) single block of 256 threads: one wave per SIMD on one CU
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(1), dim3(256), 0, 0, out, cyc, iters, 1.0);
  CK(hipDeviceSynchronize());
  long long h[4];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double c1 = (double)h[0] / (iters * inst_per_iter);
  // (b) whole chip: 256 CUs x 2 blocks of 256 threads, wall clock
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters * 4, 1.0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double flops = (double)grid * 4 * iters * 4 * inst_per_iter * flop_per_inst;
  printf("%-28s cyc/inst(1 wave/SIMD)=%7.2f  chip=%8.2f TFLOP/s (%.3f ms)\n", name, c1, flops / (ms * 1e-3) / 1e12, ms);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs=%d  clock=%d kHz  arch=%s\n", p.name, p.multiProcessorCount, p.clockRate, p.gcnArchName);
  run_rate<0>("mfma_f64_16x16x4 indep4", 2.0 * 16 * 16 * 4, 4);
  run_rate<1>("mfma_f64_16x16x4 dep", 2.0 * 16 * 16 * 4, 4);
  run_rate<2>("mfma_f64_4x4x4_4b indep4", 2.0 * 4 * 4 * 4 * 4, 4);
  run_rate<3>("mfma_f64_4x4x4_4b dep", 2.0 * 4 * 4 * 4 * 4, 4);
  run_rate<4>("v_fma_f64 indep8", 2.0 * 64, 8);
  run_rate<5>("mfma_f32_32x32x2 indep2", 2.0 * 32 * 32 * 2, 4);
  run_rate<6>("mfma_f32_16x16x4 indep4", 2.0 * 16 * 16 * 4, 4);

  double* T; CK(hipMalloc(&T, sizeof(double) * 64 * 64 * 4));
  std::vector<double> h(64 * 64 * 4);
  hipLaunchKernelGGL(probe_16x16x4, dim3(1), dim3(64), 0, 0, T);
  CK(hipMemcpy(h.data(), T, sizeof(double) * 64 * 64 * 4, hipMemcpyDeviceToHost));
  // hypothesis: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D: col=l&15, row=(l>>4)+4*r
  int bad = 0;
  for (int la = 0; la < 64; ++la) for (int lo = 0; lo < 64; ++lo) for (int r = 0; r < 4; ++r) {
    int i = la & 15, k = la >> 4, col = lo & 15, row = (lo >> 4) + 4 * r;
    double exp = (row == i) ? (double)((k << 4 | col) + 1) : 0.0;
    if (h[(la * 64 + lo) * 4 + r] != exp) ++bad;
  }
  printf("probe 16x16x4 f64: hypothesis A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15 -> mismatches=%d\n", bad);
  if (bad) { for (int la = 0; la < 64; la += 17) { printf("la=%d:", la); for (int lo = 0; lo < 64; ++lo) for (int r = 0; r < 4; ++r) if (h[(la*64+lo)*4+r] != 0) printf(" (lo%d r%d)=%g", lo, r, h[(la*64+lo)*4+r]); printf("\n"); } }
  hipLaunchKernelGGL(probe_4x4x4, dim3(1), dim3(64), 0, 0, T);
  CK(hipMemcpy(h.data(), T, sizeof(double) * 64 * 64, hipMemcpyDeviceToHost));
  printf("probe 4x4x4_4b f64 raw table: for each A lane la, list of (out lane: B lane paired)\n");
  for (int la = 0; la < 64; ++la) { printf("la=%2d:", la); for (int lo = 0; lo < 64; ++lo) if (h[la * 64 + lo] != 0) printf(" %d:%d", lo, (int)h[la * 64 + lo] - 1); printf("\n"); }
  return 0;
}
