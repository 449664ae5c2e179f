// smm_exact.h -- exact-size fp64 kernel for ONE (m, n, k) of homogeneous parameter stacks (libsmm_acc_process), compiled at run time.
//
// The reference compiles one kernel per (m, n, k) the first time a stack of that triplet arrives
// (src/acc/libsmm_acc/libsmm_acc.cpp:90-195, 281-321; five templated dataflows chosen by its autotuner's table).  Here: one
// dataflow -- the one the device-resident engine runs (mm_numeric_f64.h: cblock_f64_exact, mm_exact.h) -- on a stack: a wave owns
// `group` consecutive entries (the host sorts a stack by C offset, src/mm/dbcsr_mm_accdrv.F:481-486), keeps the sums of a run of equal
// C offsets in MFMA accumulators and adds them to C with fp64 atomics at the end of the run (runs may straddle waves), as the
// reference's kernels do (kernels/smm_acc_dnt_small.h:186-216).  Per entry: the whole A and B blocks arrive with bounds-checked
// 1 KiB buffer loads (entry s + 1 in flight in registers while entry s is multiplied from the wave's LDS slice); all piece counts,
// LDS offsets and the k loop are compile-time constants -- the run-time-size kernel (smm_stack.hip: smm_stack_f64_lds) spends ~200
// scalar / vector instructions per 23^3 product next to its 54 MFMAs.  The stack records of the group come with ONE vector load
// (lane l holds entry first + l) and are handed out with v_readlane.
//
// B is read as the host left it: transposed by libsmm_acc_transpose (n x k column-major, BT) or as stored (k x n).
// Measured (tools/acc_bench.py --threads 16, 30000-entry stacks, profiles/r06_acc_abi_threads.txt).
#ifndef DBCSR_AMD_SMM_EXACT_H
#define DBCSR_AMD_SMM_EXACT_H
#include "mm_exact.h"  // Pitch, staged_offset, cmax, LaneMap (smm_core.h), u32x4 (mm_types.h)

namespace dbcsr_amd {

// LDS bytes one wave needs; also evaluated by the host to size the launch (mm_jit.hip)
constexpr int stack_wave_lds(int m, int n, int k, bool bt) {
  const int k4 = 4 * ((k + 3) / 4);
  const int ap = m + ((m % 16 == 0) ? 2 : 0);
  const int a_lds = (ap * k4 * 8 + 15) & ~15;
  const int cb = (k * n * 8 + 1023) / 1024;
  // the B image is written in whole 1 KiB pieces; with a padded pitch (columns of 16 or 32 elements) every column is shifted by 16 bytes
  // more than the one before: at most 128 extra bytes per piece
  const int ld = bt ? n : k;
  const int bpad = (ld % 16 == 0) ? 128 : 0;
  return (a_lds + cb * (1024 + bpad) + 16 + 15) & ~15;
}

template <int M, int N, int K, bool BT>
__device__ __forceinline__ void smm_stack_exact_body(const int* __restrict__ stack, int nstack, const double* __restrict__ a_data,
                                                     const double* __restrict__ b_data, double* __restrict__ c_data, int group, char* smem) {
  constexpr int MA = (M + 7) / 8, NC = (N + 7) / 8, KS = (K + 3) / 4, K4 = 4 * KS;
  constexpr int AP = Pitch<M>::P;
  constexpr int BLD = BT ? N : K;            // leading dimension of B's image (elements per column as it lies in memory)
  constexpr int BP = Pitch<BLD>::P;
  constexpr int CA = (M * K4 * 8 + 1023) / 1024, CB = (K * N * 8 + 1023) / 1024;
  constexpr int A_LDS = (AP * K4 * 8 + 15) & ~15;
  constexpr int WAVE_LDS = stack_wave_lds(M, N, K, BT);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int first = wave * group;
  if (first >= nstack) return;
  const int cnt = min(group, nstack - first);   // <= 64: one lane per entry
  char* lds_a = smem + (size_t)wid * WAVE_LDS;
  char* lds_b = lds_a + A_LDS;
  const LaneMap L(lane);
  const int voff = lane * 16;

  // the group's records: lane l holds entry first + l
  const int* rec = stack + 3 * (size_t)(first + (lane < cnt ? lane : cnt - 1));
  const int my_a = rec[0], my_b = rec[1], my_c = rec[2];

  double acc[MA][NC];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = 0.0;
  const double* pa[MA];
  const double* pb[NC];
  const double* pbt[NC];  // last k step when K is not a multiple of 4: lanes past the end read a valid, finite element (A's padding is zero)
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    int row = 8 * a + L.rowl;
    row = row < M ? row : M - 1;
    pa[a] = reinterpret_cast<const double*>(lds_a) + row + AP * L.kq;
  }
  constexpr int kt_base = 4 * (KS - 1);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = 8 * c + L.coll;
    col = col < N ? col : N - 1;
    const int kt = kt_base + L.kq;
    if constexpr (BT) {
      pb[c] = reinterpret_cast<const double*>(lds_b) + col + BP * L.kq;
      pbt[c] = reinterpret_cast<const double*>(lds_b) + col + BP * (kt < K ? kt : 0);
    } else {
      pb[c] = reinterpret_cast<const double*>(lds_b) + L.kq + BP * col;
      pbt[c] = reinterpret_cast<const double*>(lds_b) + (kt < K ? kt : 0) + BP * col;
    }
  }
  u32x4 ra[CA], rb[CB];
  auto issue = [&](int s) {
    const int ao = __builtin_amdgcn_readlane(my_a, s), bo = __builtin_amdgcn_readlane(my_b, s);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + (ao - 1)), 0, M * K * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + (bo - 1)), 0, K * N * 8, 0x00020000);
#pragma unroll
    for (int c = 0; c < CA; ++c) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CB; ++c) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
  };
  auto flush = [&](int co) {
    double* C = c_data + (co - 1);
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
        if (row < M && col < N) unsafeAtomicAdd(C + row + M * col, acc[a][c]);
        acc[a][c] = 0.0;
      }
  };
  int cur_c = __builtin_amdgcn_readlane(my_c, 0);
  issue(0);
  for (int s = 0; s < cnt; ++s) {
    const int co = __builtin_amdgcn_readlane(my_c, s);
    if (co != cur_c) {
      flush(cur_c);
      cur_c = co;
    }
#pragma unroll
    for (int c = 0; c < CA; ++c) *reinterpret_cast<u32x4*>(lds_a + staged_offset<M>(c, lane)) = ra[c];
    DBCSR_AMD_LDS_ORDER();
#pragma unroll
    for (int c = 0; c < CB; ++c) *reinterpret_cast<u32x4*>(lds_b + staged_offset<BLD>(c, lane)) = rb[c];
    if (s + 1 < cnt) issue(s + 1);
    double av[2][MA], bv[2][NC];
    auto fetch = [&](int st, int buf) {
#pragma unroll
      for (int a = 0; a < MA; ++a) av[buf][a] = pa[a][st * 4 * AP];
#pragma unroll
      for (int c = 0; c < NC; ++c) bv[buf][c] = (st == KS - 1 && (K & 3)) ? pbt[c][0] : pb[c][BT ? st * 4 * BP : 4 * st];
    };
    fetch(0, 0);
#pragma unroll
    for (int st = 0; st < KS; ++st) {
      if (st + 1 < KS) fetch(st + 1, (st + 1) & 1);
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[st & 1][a], bv[st & 1][c], acc[a][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  flush(cur_c);
}

}  // namespace dbcsr_amd

#ifdef DBCSR_AMD_JIT_SM  // translation unit handed to hiprtc (mm_jit.hip: jit_stack_kernel): one kernel, its shape comes from the macros
#ifndef DBCSR_AMD_JIT_MINW
#define DBCSR_AMD_JIT_MINW 1
#endif
extern "C" __global__ void __launch_bounds__(256, DBCSR_AMD_JIT_MINW)
    smm_stack_f64_exact(const int* __restrict__ stack, int nstack, const double* __restrict__ a_data, const double* __restrict__ b_data,
                        double* __restrict__ c_data, int group) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  dbcsr_amd::smm_stack_exact_body<DBCSR_AMD_JIT_SM, DBCSR_AMD_JIT_SN, DBCSR_AMD_JIT_SK, (DBCSR_AMD_JIT_SBT != 0)>(stack, nstack, a_data, b_data, c_data,
                                                                                                            group, smem);
}
#endif
#endif
