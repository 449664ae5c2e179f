// mm_engine_launch.h -- part of mm_engine.hip (one translation unit; included inside namespace dbcsr_amd): the table of exact-size kernel instantiations,
// the norm kernels beside them and the launch dispatchers (block size -> template instance) of the numeric kernels.
#ifndef DBCSR_AMD_MM_ENGINE_LAUNCH_H
#define DBCSR_AMD_MM_ENGINE_LAUNCH_H

// ----------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------
// Exact-size kernels are instantiated for every cube from 9 to 32 (the reference compiles one kernel per (m, n, k) at run
// time; here the list is fixed at build time and every other case -- mixed sizes, blocks above 32 -- runs the generic kernels; measured on 4 x 4 blocks the generic kernel is 7 % faster, so sizes
// up to 8 are left to it).
#define DBCSR_AMD_HOT_SIZES(X) \
  X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)

#define DBCSR_AMD_DMA_SIZES(X) X(13) X(16) X(23) X(32)

// one wave per C block, only the blocks that are NOT m x n: their squared Frobenius norm (the exact-size kernel wrote the others')
__global__ void __launch_bounds__(256) block_norms_other_sizes(const Desc* __restrict__ descs, int64_t nblk, const double* __restrict__ c_data,
                                                               int m, int n, double* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t cb = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (cb >= nblk) return;
  const Desc d = descs[cb];
  if (d.m == m && d.n == n) return;
  const double* x = c_data + d.c_off;
  double ss = 0.0;
  for (int e = lane; e < d.m * d.n; e += 64) ss += x[e] * x[e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
  if (lane == 0) norms[cb] = ss;
}

// the same for a multiply of mixed sizes: the blocks whose (m, n) class had no run-time compiled kernel (class 9 = other sizes, or hiprtc failed)
struct ClassSet {
  int m[3], n[3], jit_mask;
};
__global__ void __launch_bounds__(256) block_norms_unserved_classes(const Desc* __restrict__ descs, int64_t nblk, const double* __restrict__ c_data,
                                                                    ClassSet cs, double* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t cb = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (cb >= nblk) return;
  const Desc d = descs[cb];
  int rm = 3, rn = 3;
#pragma unroll
  for (int q = 2; q >= 0; --q) {
    if (cs.m[q] > 0 && d.m == cs.m[q]) rm = q;
    if (cs.n[q] > 0 && d.n == cs.n[q]) rn = q;
  }
  if (rm < 3 && rn < 3 && ((cs.jit_mask >> (3 * rm + rn)) & 1)) return;  // its class kernel wrote the norm
  const double* x = c_data + d.c_off;
  double ss = 0.0;
  for (int e = lane; e < d.m * d.n; e += 64) ss += x[e] * x[e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
  if (lane == 0) norms[cb] = ss;
}

static bool launch_hot_f64(int m, int n, int k, dim3 grid, size_t lds_bytes, hipStream_t st, const Desc* descs, int64_t nblk,
                           const Entry* entries, const double* a_data, const double* b_data, double* c_out, const double* c_in,
                           double alpha, double beta, int lds_a, int lds_wave, int dbg, const int* order, const Work* work, int wg_waves,
                           double* norms, int variant) {
  if (m != n || m != k) return false;
#ifdef DBCSR_AMD_EXPERIMENTS
  // profiling variants exist for the benchmark's block size only (ablation switches; unpaired fragment reads)
  if (m == 23 && variant >= 1 && variant <= 6) {
#define DBCSR_HOT_VARIANT(V_)                                                                                                              \
  hipLaunchKernelGGL((mm_numeric_f64_hot<23, 23, 23, V_>), grid, dim3(64 * wg_waves), lds_bytes, st, descs, nblk, entries, a_data, b_data, c_out, \
                     c_in, alpha, beta, lds_a, lds_wave, dbg, order, work, norms)
    switch (variant) {
      case 1: DBCSR_HOT_VARIANT(1); break;
      case 2: DBCSR_HOT_VARIANT(2); break;
      case 3: DBCSR_HOT_VARIANT(3); break;
      case 4: DBCSR_HOT_VARIANT(4); break;
      case 5: DBCSR_HOT_VARIANT(5); break;
      default: DBCSR_HOT_VARIANT(6); break;
    }
#undef DBCSR_HOT_VARIANT
    return true;
  }
#else
  (void)variant;
#endif
  switch (m) {
#define DBCSR_HOT_CASE(S_)                                                                                                      \
  case S_:                                                                                                                      \
    hipLaunchKernelGGL((mm_numeric_f64_hot<S_, S_, S_, 0>), grid, dim3(64 * wg_waves), lds_bytes, st, descs, nblk, entries, a_data, b_data, c_out, \
                       c_in, alpha, beta, lds_a, lds_wave, dbg, order, work, norms);                                            \
    return true;
    DBCSR_AMD_HOT_SIZES(DBCSR_HOT_CASE)
#undef DBCSR_HOT_CASE
    default: return false;
  }
}

#ifdef DBCSR_AMD_EXPERIMENTS
// LDS-DMA variant of the exact-size kernel (mm_dma.h): S ring slots per wave, one wave per workgroup
template <int S_>
static bool launch_dma_f64_s(int m, int n, int k, unsigned npos, hipStream_t st, const Desc* descs, int64_t nblk, const Entry* entries,
                             const double* a_data, const double* b_data, double* c_out, const double* c_in, double alpha, double beta,
                             int skip_empty, const int* order) {
  if (m != n || m != k) return false;
  switch (m) {
#define DBCSR_DMA_CASE(S__)                                                                                                   \
  case S__:                                                                                                                   \
    hipLaunchKernelGGL((mm_numeric_f64_dma<S__, S__, S__, S_>), dim3(npos), dim3(64), (DmaRing<S__, S__, S__, S_>::BYTES), st, descs, \
                       nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order);                           \
    return true;
    DBCSR_AMD_DMA_SIZES(DBCSR_DMA_CASE)
#undef DBCSR_DMA_CASE
    default: return false;
  }
}
static bool launch_dma_f64(int S, int m, int n, int k, unsigned npos, hipStream_t st, const Desc* descs, int64_t nblk,
                           const Entry* entries, const double* a_data, const double* b_data, double* c_out, const double* c_in,
                           double alpha, double beta, int skip_empty, const int* order) {
  switch (S) {
    case 2: return launch_dma_f64_s<2>(m, n, k, npos, st, descs, nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order);
    case 3: return launch_dma_f64_s<3>(m, n, k, npos, st, descs, nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order);
    case 4: return launch_dma_f64_s<4>(m, n, k, npos, st, descs, nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order);
    default: return false;
  }
}
#endif

static bool launch_hot_f32(int m, int n, int k, dim3 grid, int wg_waves, hipStream_t st, const Desc* descs, int64_t nblk, const Entry* entries,
                           const float* a_data, const float* b_data, float* c_out, const float* c_in, float alpha, float beta,
                           int skip_empty, const int* order) {
  if (m != n || m != k) return false;
  switch (m) {
#define DBCSR_HOT_CASE(S_)                                                                                                     \
  case S_:                                                                                                                     \
    hipLaunchKernelGGL((mm_numeric_f32_hot<S_, S_, S_>), grid, dim3(64 * wg_waves), f32_lds_bytes(wg_waves), st, descs, nblk, entries, a_data, b_data, c_out, c_in, \
                       alpha, beta, skip_empty, order);                                                                        \
    return true;
    DBCSR_AMD_HOT_SIZES(DBCSR_HOT_CASE)
#undef DBCSR_HOT_CASE
    default: return false;
  }
}

// the direct form of the fp32 exact-size kernel (mm_numeric_f32.h, round 5): cubes whose k is a multiple of 8
static bool launch_hot_f32_direct(int m, int n, int k, dim3 grid, int wg_waves, hipStream_t st, const Desc* descs, int64_t nblk, const Entry* entries,
                                  const float* a_data, const float* b_data, float* c_out, const float* c_in, float alpha, float beta,
                                  int skip_empty, const int* order, bool slim = false) {
  if (m != n || m != k) return false;
  if (slim) {   // every C block has the dominant size: LDS for the B images only
    switch (m) {
#define DBCSR_SLIM_CASE(S_)                                                                                                    \
  case S_:                                                                                                                     \
    hipLaunchKernelGGL((mm_numeric_f32_direct_slim<S_, S_, S_>), grid, dim3(64 * wg_waves), (size_t)wg_waves * f32d_wave_floats(S_) * sizeof(float), st, \
                       descs, nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order);                     \
    return true;
      DBCSR_SLIM_CASE(16) DBCSR_SLIM_CASE(24) DBCSR_SLIM_CASE(32)
#undef DBCSR_SLIM_CASE
      default: return false;
    }
  }
  switch (m) {
#define DBCSR_DIRECT_CASE(S_)                                                                                                  \
  case S_:                                                                                                                     \
    hipLaunchKernelGGL((mm_numeric_f32_direct<S_, S_, S_>), grid, dim3(64 * wg_waves), f32_lds_bytes(wg_waves), st, descs, nblk, entries, a_data, b_data, \
                       c_out, c_in, alpha, beta, skip_empty, order);                                                           \
    return true;
    DBCSR_DIRECT_CASE(16) DBCSR_DIRECT_CASE(24) DBCSR_DIRECT_CASE(32)
#undef DBCSR_DIRECT_CASE
    default: return false;
  }
}

// blocks of 33 ... 80: sub-blocks of TM x TN tiles per wave, 2 x 2 waves per C block (mm_numeric_f64_big.h)
static bool launch_big_f64(int tm, int tn, unsigned npos, hipStream_t st, const Desc* descs, int64_t nblk, const Entry* entries, const double* a_data,
                           const double* b_data, double* c_out, const double* c_in, double alpha, double beta, int skip_empty, const int* order) {
  if (tm < 2 || tm > 5 || tn < 2 || tn > 5 || npos == 0) return false;
  switch (tm * 8 + tn) {
#define DBCSR_BIG_CASE(A_, B_)                                                                                                        \
  case A_ * 8 + B_:                                                                                                                   \
    hipLaunchKernelGGL((mm_numeric_f64_big<A_, B_>), dim3(npos), dim3(256), (size_t)big_lds_bytes(A_, B_), st, descs, nblk, entries, a_data, b_data, c_out, \
                       c_in, alpha, beta, skip_empty, order);                                                                         \
    return true;
    DBCSR_BIG_CASE(2, 2) DBCSR_BIG_CASE(2, 3) DBCSR_BIG_CASE(2, 4) DBCSR_BIG_CASE(2, 5)
    DBCSR_BIG_CASE(3, 2) DBCSR_BIG_CASE(3, 3) DBCSR_BIG_CASE(3, 4) DBCSR_BIG_CASE(3, 5)
    DBCSR_BIG_CASE(4, 2) DBCSR_BIG_CASE(4, 3) DBCSR_BIG_CASE(4, 4) DBCSR_BIG_CASE(4, 5)
    DBCSR_BIG_CASE(5, 2) DBCSR_BIG_CASE(5, 3) DBCSR_BIG_CASE(5, 4) DBCSR_BIG_CASE(5, 5)
#undef DBCSR_BIG_CASE
    default: return false;
  }
}

#endif
