// smm_stack.hip -- libsmm_acc_process / libsmm_acc_transpose / c_calculate_norms
// of the DBCSR accelerator C-ABI (include/dbcsr_acc_libsmm.h) as hand-written
// gfx950 kernels.  Replaces /root/reference/src/acc/libsmm_acc/libsmm_acc.cpp
// (run-time JIT of five templated CUDA/HIP kernels) and
// src/acc/cuda_hip/calculate_norms.cpp.
//
// Stack kernel: one wavefront per group of consecutive stack entries.  The host
// (src/mm/dbcsr_mm_accdrv.F:481-486) sorts a stack by C offset, so a wave keeps
// the C block in MFMA accumulators across a run of equal c and adds it to
// memory once per run with hardware fp64/fp32 atomics (runs may straddle
// waves, and several host threads' stacks never share C blocks).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "../../include/dbcsr_acc_libsmm.h"
#include <algorithm>

#include "common.h"
#include "mm_types.h"
#include "smm_core.h"
#include "mm_numeric_f64_big.h"   // slab geometry of the workgroup-per-C-block kernel (BIG_KSL, BIG_PB, big_*_bytes)
#include "mm_numeric_f64_mid.h"   // BigSub: a wave's block covered in units of 4 x 4 (the one-wave slab kernels)
#include "mm_jit.h"               // jit_stack_kernel: the exact-size kernel of smm_exact.h, compiled per (m, n, k) at run time

namespace dbcsr_amd {

constexpr int kStackGroup = 16;  // default number of stack entries per wavefront (see stack_group())

// name of the kernel the calling host thread's last libsmm_acc_process launched (dbcsr_amd_smm_last_kernel: tests and acc_bench say
// which dataflow they measured)
thread_local char g_last_smm_kernel[64] = "";
static void note_kernel(const char* fmt, int m, int n, int k, bool bt) { snprintf(g_last_smm_kernel, sizeof g_last_smm_kernel, fmt, m, n, k, bt ? "; B transposed" : ""); }

template <int MA, int NC, bool BT>
__global__ void __launch_bounds__(256) smm_stack_f64(const int* __restrict__ stack, int nstack, const double* __restrict__ a_data,
                                                     const double* __restrict__ b_data, double* __restrict__ c_data, int m,
                                                     int n, int k, int group) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int row0 = blockIdx.y * (8 * MA), col0 = blockIdx.z * (8 * NC);
  const int first = wave * group;
  if (first >= nstack) return;
  const int last = min(first + group, nstack);
  const LaneMap L(lane);
  double acc[MA][NC];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = 0.0;
  int cur_c = __builtin_amdgcn_readfirstlane(stack[3 * first + 2]);
  for (int s = first; s <= last; ++s) {
    int ao = 0, bo = 0, co = -1;
    if (s < last) {
      ao = __builtin_amdgcn_readfirstlane(stack[3 * s]);
      bo = __builtin_amdgcn_readfirstlane(stack[3 * s + 1]);
      co = __builtin_amdgcn_readfirstlane(stack[3 * s + 2]);
    }
    if (co != cur_c) {  // end of a run of equal C offsets: C += acc (1-based offsets)
      double* C = c_data + (cur_c - 1);
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int row = row0 + 8 * a + L.rowd, col = col0 + 8 * c + L.coll;
          if (row < m && col < n) unsafeAtomicAdd(C + row + (size_t)m * col, acc[a][c]);
          acc[a][c] = 0.0;
        }
      cur_c = co;
    }
    if (s < last) block_product_f64<MA, NC, BT>(acc, a_data + (ao - 1), b_data + (bo - 1), m, n, k, L, row0, col0);
  }
}


// LDS-staged variant for blocks up to 32 x 32 (same staging as the device-resident engine, mm_engine.hip):
// whole A and B blocks of a stack entry are copied with bounds-checked 1 KiB buffer loads into the wave's private
// LDS slice, the next entry's blocks are in flight in registers meanwhile.
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

template <int MA, int NC, bool BT>
__global__ void __launch_bounds__(256) smm_stack_f64_lds(const int* __restrict__ stack, int nstack, const double* __restrict__ a_data,
                                                         const double* __restrict__ b_data, double* __restrict__ c_data, int m,
                                                         int n, int k, int group, int lds_a_bytes, int lds_wave_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CA = 2 * MA, CB = 2 * NC;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int first = wave * group;
  if (first >= nstack) return;
  const int last = min(first + group, nstack);
  char* lds_a = smem + (size_t)wid * lds_wave_bytes;
  char* lds_b = lds_a + lds_a_bytes;
  const LaneMap L(lane);
  const int voff = lane * 16;
  const int abytes = m * k * 8, bbytes = k * n * 8;
  const int nca = (m * ((k + 3) & ~3) * 8 + 1023) >> 10, ncb = (bbytes + 1023) >> 10;
  double acc[MA][NC];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = 0.0;
  u32x4s ra[CA], rb[CB];
  auto issue = [&](int s) {
    const int ao = __builtin_amdgcn_readfirstlane(stack[3 * s]), bo = __builtin_amdgcn_readfirstlane(stack[3 * s + 1]);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + (ao - 1)), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + (bo - 1)), 0, bbytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < CA; ++c)
      if (c < nca) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CB; ++c)
      if (c < ncb) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
  };
  auto flush = [&](int co) {
    double* C = c_data + (co - 1);
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
        if (row < m && col < n) unsafeAtomicAdd(C + row + (size_t)m * col, acc[a][c]);
        acc[a][c] = 0.0;
      }
  };
  int cur_c = __builtin_amdgcn_readfirstlane(stack[3 * first + 2]);
  issue(first);
  for (int s = first; s < last; ++s) {
    const int co = __builtin_amdgcn_readfirstlane(stack[3 * s + 2]);
    if (co != cur_c) {
      flush(cur_c);
      cur_c = co;
    }
#pragma unroll
    for (int c = 0; c < CA; ++c)
      if (c < nca) *reinterpret_cast<u32x4s*>(lds_a + c * 1024 + voff) = ra[c];
#pragma unroll
    for (int c = 0; c < CB; ++c)
      if (c < ncb) *reinterpret_cast<u32x4s*>(lds_b + c * 1024 + voff) = rb[c];
    if (s + 1 < last) issue(s + 1);
    block_product_f64_lds<MA, NC, BT>(acc, reinterpret_cast<const double*>(lds_a), reinterpret_cast<const double*>(lds_b), m, n, k, L);
  }
  flush(cur_c);
}

// Blocks of 33 ... 80 (or an inner dimension above 32) under the acc ABI (round 5): the dataflow of mm_numeric_f64_big.h on a parameter
// stack.  A WORKGROUP takes `group` consecutive stack entries; its four waves own the C block as 2 x 2 sub-blocks of TM x TN MFMA tiles
// and keep the sums in registers across runs of equal C offsets (atomic adds at the end of a run, as the other stack kernels and the
// reference's kernels do, kernels/smm_acc_dnt_largeDB2.h:159-314); a product is consumed in slabs of 16 inner indices copied once into
// LDS by all 256 threads and read by every wave, double-buffered over slabs AND entries.  B as libsmm_acc_transpose leaves it (n x k, BT):
// its slab is as contiguous as A's; B as stored (k x n): n runs of 128 bytes with the padded pitch of the engine's kernel.
// (smm_stack_f64 walks 32 x 32 tiles one after the other with fragments from global memory: 11.6 TFLOP/s at 72^3, acc_bench.)
template <int TM, int TN, bool BT, bool EXACT>
__global__ void __launch_bounds__(256) smm_stack_f64_big(const int* __restrict__ stack, int nstack, const double* __restrict__ a_data,
                                                         const double* __restrict__ b_data, double* __restrict__ c_data, int m, int n, int k,
                                                         int group_arg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int group = group_arg & 0xffff;
  const bool full_tiles = (group_arg >> 16) & 1;   // DBCSR_AMD_SMM_BIG_EXACT=0 (measurements): every wave issues all TM x TN tile products
  constexpr int RA = big_a_rounds(TM), RBC = big_a_rounds(TN), RBS = big_b_rounds(TN), RB = BT ? RBC : RBS;
  constexpr int ABYTES = big_a_bytes(TM), BUF = ABYTES + (BT ? big_a_bytes(TN) : big_b_bytes(TN));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int first = blockIdx.x * group;
  if (first >= nstack) return;
  const int last = min(first + group, nstack);
  const LaneMap L(lane);
  const int mt = (m + 7) >> 3, nt = (n + 7) >> 3;
  const int ta0 = (wid >> 1) * ((mt + 1) >> 1), tc0 = (wid & 1) * ((nt + 1) >> 1);
  double acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int c = 0; c < TN; ++c) acc[a][c] = 0.0;
  int fa[TM], fb[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    int row = 8 * (ta0 + a) + L.rowl;
    row = row < m ? row : m - 1;
    fa[a] = row + m * L.kq;
  }
#pragma unroll
  for (int c = 0; c < TN; ++c) {
    int col = 8 * (tc0 + c) + L.coll;
    col = col < n ? col : n - 1;
    fb[c] = ABYTES / 8 + (BT ? col + n * L.kq : col * BIG_PB + L.kq);
  }
  const int astep = 4 * m, bstep = BT ? 4 * n : 4;
  const int bk = 2 * (tid & 7), bc = tid >> 3;
  u32x4 ga[RA], gb[RB];
  int k0_cur = 0;
  auto issue = [&](int s, int k0) __attribute__((always_inline)) {
    const int ao = __builtin_amdgcn_readfirstlane(stack[3 * s]) - 1, bo = __builtin_amdgcn_readfirstlane(stack[3 * s + 1]) - 1;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + ao), 0, m * k * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + bo), 0, k * n * 8, 0x00020000);
    const int abase = tid * 16 + k0 * m * 8;   // (the whole offset in the bounds-checked operand: the k tail must arrive as zeros)
#pragma unroll
    for (int r = 0; r < RA; ++r) ga[r] = __builtin_amdgcn_raw_buffer_load_b128(rsa, abase + r * 4096, 0, 0);
    if constexpr (BT) {
      const int bbase = tid * 16 + k0 * n * 8;
#pragma unroll
      for (int r = 0; r < RB; ++r) gb[r] = __builtin_amdgcn_raw_buffer_load_b128(rsb, bbase + r * 4096, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int col = 32 * r + bc;
        const int off = col < n ? (col * k + k0 + bk) * 8 : 0x7ffffff0;
        gb[r] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off, 0, 0);
      }
    }
    k0_cur = k0;
  };
  auto stage = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < RA; ++r) *reinterpret_cast<u32x4*>(buf + r * 4096 + tid * 16) = ga[r];
    if constexpr (BT) {
#pragma unroll
      for (int r = 0; r < RB; ++r) *reinterpret_cast<u32x4*>(buf + ABYTES + r * 4096 + tid * 16) = gb[r];
    } else {
      const bool k0ok = k0_cur + bk < k, k1ok = k0_cur + bk + 1 < k;
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        u32x4 v = gb[r];
        if (!k0ok) v[0] = 0u, v[1] = 0u;
        if (!k1ok) v[2] = 0u, v[3] = 0u;
        if (32 * r + bc < 16 * TN) *reinterpret_cast<u32x4*>(buf + ABYTES + ((32 * r + bc) * BIG_PB + bk) * 8) = v;
      }
    }
  };
  auto flush = [&](int co) __attribute__((always_inline)) {
    double* C = c_data + (co - 1);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int c = 0; c < TN; ++c) {
        const int ta = ta0 + a, tc = tc0 + c;
        const int row = 8 * ta + L.rowd, col = 8 * tc + L.coll;
        const bool mine = ta < ((wid >> 1) ? mt : ((mt + 1) >> 1)) && tc < ((wid & 1) ? nt : ((nt + 1) >> 1));
        if (mine && row < m && col < n) unsafeAtomicAdd(C + row + (size_t)m * col, acc[a][c]);
        acc[a][c] = 0.0;
      }
  };
  int s = first, k0 = 0, it = 0;
  int cur_c = __builtin_amdgcn_readfirstlane(stack[3 * first + 2]);
  issue(first, 0);
  // every wave multiplies exactly the TA x TC tiles it owns (see mm_numeric_f64_big.h: the variant is chosen once, outside the loop)
  auto run = [&](auto ta_tag, auto tc_tag) __attribute__((always_inline)) {
    constexpr int TA = decltype(ta_tag)::value, TC = decltype(tc_tag)::value;
    while (s < last) {
      char* buf = smem + (it & 1) * BUF;
      if (k0 == 0) {   // a new entry: the end of a run of equal C offsets?
        const int co = __builtin_amdgcn_readfirstlane(stack[3 * s + 2]);
        if (co != cur_c) {
          flush(cur_c);
          cur_c = co;
        }
      }
      const int rem = (k - k0 + 3) >> 2;
      const int nst = rem < BIG_KSL / 4 ? rem : BIG_KSL / 4;
      stage(buf);
      __syncthreads();
      int s2 = s, k2 = k0 + BIG_KSL;
      if (k2 >= k) s2 = s + 1, k2 = 0;
      if (s2 < last) issue(s2, k2);
      const double* la = reinterpret_cast<const double*>(buf);
      double av[2][TA], bv[2][TC];
      auto fetch = [&](int st_, int set) {
#pragma unroll
        for (int a = 0; a < TA; ++a) av[set][a] = la[fa[a] + astep * st_];
#pragma unroll
        for (int c = 0; c < TC; ++c) bv[set][c] = la[fb[c] + bstep * st_];
      };
      fetch(0, 0);
#pragma unroll
      for (int st_ = 0; st_ < BIG_KSL / 4; ++st_) {
        if (st_ + 1 < nst) fetch(st_ + 1, (st_ + 1) & 1);
        if (st_ < nst) {
#pragma unroll
          for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int c = 0; c < TC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[st_ & 1][a], bv[st_ & 1][c], acc[a][c], 0, 0, 0);
        }
      }
      s = s2;
      k0 = k2;
      ++it;
    }
    flush(cur_c);   // (inside the variant: accumulators that met again behind the switch were kept twice)
  };
  {
    const int own_r = (wid >> 1) ? mt - ((mt + 1) >> 1) : ((mt + 1) >> 1), own_c = (wid & 1) ? nt - ((nt + 1) >> 1) : ((nt + 1) >> 1);
    if constexpr (EXACT) {
      switch (full_tiles ? 0 : (own_r >= TM ? 0 : 2) + (own_c >= TN ? 0 : 1)) {   // (wave-uniform)
        case 0: run(std::integral_constant<int, TM>{}, std::integral_constant<int, TN>{}); break;
        case 1: run(std::integral_constant<int, TM>{}, std::integral_constant<int, TN - 1>{}); break;
        case 2: run(std::integral_constant<int, TM - 1>{}, std::integral_constant<int, TN>{}); break;
        default: run(std::integral_constant<int, TM - 1>{}, std::integral_constant<int, TN - 1>{}); break;
      }
    } else {   // even tile counts in both dimensions: the four sub-blocks are equally large, one variant (and fewer registers: three waves per SIMD for 5 x 5)
      (void)own_r;
      (void)own_c;
      run(std::integral_constant<int, TM>{}, std::integral_constant<int, TN>{});
    }
  }
}

template <bool BT>
__global__ void __launch_bounds__(256) smm_stack_f32(const int* __restrict__ stack, int nstack, const float* __restrict__ a_data,
                                                     const float* __restrict__ b_data, float* __restrict__ c_data, int m, int n,
                                                     int k, int group) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int row0 = blockIdx.y * 32, col0 = blockIdx.z * 32;
  const int first = wave * group;
  if (first >= nstack) return;
  const int last = min(first + group, nstack);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  int cur_c = __builtin_amdgcn_readfirstlane(stack[3 * first + 2]);
  for (int s = first; s <= last; ++s) {
    int ao = 0, bo = 0, co = -1;
    if (s < last) {
      ao = __builtin_amdgcn_readfirstlane(stack[3 * s]);
      bo = __builtin_amdgcn_readfirstlane(stack[3 * s + 1]);
      co = __builtin_amdgcn_readfirstlane(stack[3 * s + 2]);
    }
    if (co != cur_c) {
      float* C = c_data + (cur_c - 1);
      const int col = col0 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < m && col < n) unsafeAtomicAdd(C + row + (size_t)m * col, acc[r]);
        acc[r] = 0.0f;
      }
      cur_c = co;
    }
    if (s < last) block_product_f32<BT>(acc, a_data + (ao - 1), b_data + (bo - 1), m, n, k, lane, row0, col0);
  }
}


// fp32, blocks up to 32 x 32, LDS-staged: A copied as stored, B written transposed with a 33-float pitch (see
// mm_numeric_f32_lds in mm_engine.hip for the reasoning).  fp32 stacks always carry B as stored (k x n): the
// reference's transpose pass is a no-op for fp32 (libsmm_acc.cpp:484).
__global__ void __launch_bounds__(256) smm_stack_f32_lds(const int* __restrict__ stack, int nstack, const float* __restrict__ a_data,
                                                         const float* __restrict__ b_data, float* __restrict__ c_data, int m, int n,
                                                         int k, int group) {
  constexpr int CH = 4, LDN = 33, A_FLOATS = 1024 + 64, BT_FLOATS = ((LDN * 32 + 31) + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float smem[4 * (A_FLOATS + BT_FLOATS)];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int first = wave * group;
  if (first >= nstack) return;
  const int last = min(first + group, nstack);
  float* lds_a = smem + (size_t)wid * (A_FLOATS + BT_FLOATS);
  float* lds_bt = lds_a + A_FLOATS;
  const int voff = lane * 16;
  const int kn = k * n;
  const int nca = (m * (k + 1) * 4 + 1023) >> 10, ncb = (kn * 4 + 1023) >> 10;
  const unsigned inv = (65536u + (unsigned)k - 1u) / (unsigned)k;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  u32x4s ra[CH], rb[CH];
  auto issue = [&](int s) {
    const int ao = __builtin_amdgcn_readfirstlane(stack[3 * s]), bo = __builtin_amdgcn_readfirstlane(stack[3 * s + 1]);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + (ao - 1)), 0, m * k * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + (bo - 1)), 0, kn * 4, 0x00020000);
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < nca) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < ncb) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
  };
  const int i = lane & 31, kh = lane >> 5;
  const int arow = i < m ? i : m - 1, bcol = i < n ? i : n - 1;
  auto flush = [&](int co) {
    float* C = c_data + (co - 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < m && i < n) unsafeAtomicAdd(C + row + (size_t)m * i, acc[r]);
      acc[r] = 0.0f;
    }
  };
  int cur_c = __builtin_amdgcn_readfirstlane(stack[3 * first + 2]);
  issue(first);
  for (int s = first; s < last; ++s) {
    const int co = __builtin_amdgcn_readfirstlane(stack[3 * s + 2]);
    if (co != cur_c) {
      flush(cur_c);
      cur_c = co;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < nca) *reinterpret_cast<u32x4s*>(reinterpret_cast<char*>(lds_a) + c * 1024 + voff) = ra[c];
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < ncb) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const unsigned el = (unsigned)((c * 64 + lane) * 4 + t);
          const unsigned j = (el * inv) >> 16, kk = el - j * (unsigned)k;
          if ((int)el < kn) lds_bt[j + LDN * kk] = __uint_as_float(rb[c][t]);
        }
      }
    if (s + 1 < last) issue(s + 1);
    const int nsteps = (k + 1) >> 1;
    int aoff = arow + m * kh;
    for (int s2 = 0; s2 < nsteps; ++s2) {
      const int kk = 2 * s2 + kh;
      const float av = lds_a[aoff];
      const float bv = lds_bt[bcol + LDN * (kk < k ? kk : k - 1)];
      aoff += 2 * m;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  flush(cur_c);
}

// In-place transpose of listed m x n column-major blocks (-> n x m column-major).
// One workgroup per block, the block staged in LDS.
__global__ void __launch_bounds__(256) transpose_blocks_f64(const int* __restrict__ trs_stack, double* __restrict__ data, int m,
                                                            int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* buf = reinterpret_cast<double*>(smem);
  double* blk = data + trs_stack[blockIdx.x];
  const int mn = m * n;
  for (int i = threadIdx.x; i < mn; i += blockDim.x) buf[i] = blk[i];
  __syncthreads();
  for (int i = threadIdx.x; i < mn; i += blockDim.x) {
    const int r = i % n, c = i / n;  // element (r, c) of the n x m result = element (c, r) of the source
    blk[i] = buf[r * m + c];
  }
}

// norms[b] = sum_j mat[offsets[b] + j]^2 ; one wavefront per block.
__global__ void __launch_bounds__(256) block_norms_f64(const double* __restrict__ mat, int nblks, const int* __restrict__ offsets,
                                                       const int* __restrict__ nelems, float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (b >= nblks) return;
  const double* p = mat + offsets[b];
  const int ne = nelems[b];
  double s = 0.0;
  for (int i = lane; i < ne; i += 64) s += p[i] * p[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) norms[b] = (float)s;
}

// Stack entries per wavefront.  A wave keeps the C block of a run of equal c offsets in its accumulators, so long groups
// save atomics and wave start-ups.  Measured with tools/stack_group_sweep.sh (16005-entry stacks, groups 2/4/8/16):
// 23^3 6.5/9.2/12.4/12.7 TFLOP/s, 32^3 14.8/18.6/21.7/21.7, 13^3 3.3/4.7/5.7/4.5, 5^3 (30000 entries) 0.92/0.95/0.92/0.76
// -> 16 entries for blocks of at least 8000 multiply-adds, 8 below.  DBCSR_AMD_STACK_GROUP overrides (experiments).
static int stack_group(int m, int n, int k) {
  static const int forced = getenv("DBCSR_AMD_STACK_GROUP") ? atoi(getenv("DBCSR_AMD_STACK_GROUP")) : 0;
  if (forced > 0) return forced;
  return (int64_t)m * n * k >= 8000 ? kStackGroup : kStackGroup / 2;
}

template <int MA, int NC>
static int launch_f64(bool bt, dim3 grid, hipStream_t st, const int* stack, int nstack, const double* a, const double* b, double* c,
                      int m, int n, int k, int group) {
  if (m <= 32 && n <= 32 && k <= 32 && grid.y == 1 && grid.z == 1) {  // LDS-staged kernel (whole blocks fit the staging chunks)
    const int lds_a = ((m * ((k + 3) & ~3) * 8 + 1023) / 1024) * 1024, lds_b = ((k * n * 8 + 1023) / 1024) * 1024;
    const size_t lds = (size_t)4 * (lds_a + lds_b);
    if (bt)
      hipLaunchKernelGGL((smm_stack_f64_lds<MA, NC, true>), grid, dim3(256), lds, st, stack, nstack, a, b, c, m, n, k, group, lds_a, lds_a + lds_b);
    else
      hipLaunchKernelGGL((smm_stack_f64_lds<MA, NC, false>), grid, dim3(256), lds, st, stack, nstack, a, b, c, m, n, k, group, lds_a, lds_a + lds_b);
    return dbcsr_amd::check(hipGetLastError(), "smm_stack_f64_lds launch", __FILE__, __LINE__);
  }
  if (bt)
    hipLaunchKernelGGL((smm_stack_f64<MA, NC, true>), grid, dim3(256), 0, st, stack, nstack, a, b, c, m, n, k, group);
  else
    hipLaunchKernelGGL((smm_stack_f64<MA, NC, false>), grid, dim3(256), 0, st, stack, nstack, a, b, c, m, n, k, group);
  return dbcsr_amd::check(hipGetLastError(), "smm_stack_f64 launch", __FILE__, __LINE__);
}

template <int MA>
static int launch_f64_nc(int NC, bool bt, dim3 grid, hipStream_t st, const int* stack, int nstack, const double* a,
                         const double* b, double* c, int m, int n, int k, int group) {
  switch (NC) {
    case 1: return launch_f64<MA, 1>(bt, grid, st, stack, nstack, a, b, c, m, n, k, group);
    case 2: return launch_f64<MA, 2>(bt, grid, st, stack, nstack, a, b, c, m, n, k, group);
    case 3: return launch_f64<MA, 3>(bt, grid, st, stack, nstack, a, b, c, m, n, k, group);
    default: return launch_f64<MA, 4>(bt, grid, st, stack, nstack, a, b, c, m, n, k, group);
  }
}

template <int TM, int TN>
static int launch_f64_big(bool bt, hipStream_t st, const int* stack, int nstack, const double* a, const double* b, double* c, int m, int n, int k) {
  // stack entries per workgroup: sums are kept in registers across the entries of a run of equal C offsets and flushed with atomic adds at its
  // end (and at the workgroup's last entry), so longer groups flush less often but leave fewer workgroups to balance (DBCSR_AMD_SMM_BIG_GROUP)
  static const int group_env = getenv("DBCSR_AMD_SMM_BIG_GROUP") ? atoi(getenv("DBCSR_AMD_SMM_BIG_GROUP")) : 0;
  const int group8 = group_env > 0 && group_env < 65536 ? group_env : 8;
  static const int full = (getenv("DBCSR_AMD_SMM_BIG_EXACT") != nullptr && atoi(getenv("DBCSR_AMD_SMM_BIG_EXACT")) == 0) ? 1 : 0;
  const int group = group8 | (full << 16);
  const dim3 grid((unsigned)((nstack + group8 - 1) / group8));
  // a stack is homogeneous: whether the 2 x 2 sub-blocks are equally large is known here
  const int mt = (m + 7) / 8, nt = (n + 7) / 8;
  const bool even = mt == 2 * TM && nt == 2 * TN;   // every wave owns exactly TM x TN tiles
  const size_t lds_t = (size_t)2 * (big_a_bytes(TM) + big_a_bytes(TN)), lds_s = (size_t)2 * (big_a_bytes(TM) + big_b_bytes(TN));
  if (bt && even)
    hipLaunchKernelGGL((smm_stack_f64_big<TM, TN, true, false>), grid, dim3(256), lds_t, st, stack, nstack, a, b, c, m, n, k, group);
  else if (bt)
    hipLaunchKernelGGL((smm_stack_f64_big<TM, TN, true, true>), grid, dim3(256), lds_t, st, stack, nstack, a, b, c, m, n, k, group);
  else if (even)
    hipLaunchKernelGGL((smm_stack_f64_big<TM, TN, false, false>), grid, dim3(256), lds_s, st, stack, nstack, a, b, c, m, n, k, group);
  else
    hipLaunchKernelGGL((smm_stack_f64_big<TM, TN, false, true>), grid, dim3(256), lds_s, st, stack, nstack, a, b, c, m, n, k, group);
  return dbcsr_amd::check(hipGetLastError(), "smm_stack_f64_big launch", __FILE__, __LINE__);
}

// Blocks of 33 ... 40 in both dimensions under the acc ABI (round 6): the one-wave slab dataflow of mm_numeric_f64_mid.h on a parameter stack.  A WAVE
// (a workgroup of 64 threads) takes `group` consecutive entries, owns the whole C block in units of 4 x 4 (BigSub<RBX, CBX>: 21 MFMAs per k step for
// 33 ... 36 instead of the 25 of a block padded to 40), keeps the sums across runs of equal C offsets and adds them to C with fp64 atomics at the end of a
// run; an entry's operands arrive in slabs of 8 inner indices through 5-6 KB of LDS, the next slab in flight in registers.  B as libsmm_acc_transpose
// leaves it (n x k: its slab is as contiguous as A's) or as stored (k x n: n runs of 64 bytes at a pitch of 12 doubles).  The workgroup kernel above is
// LDS-bound at these sizes (profiles/r06_big_blocks_sub4_experiment.txt, r06_mid_blocks.txt).
template <int RBX, int CBX, bool BT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) smm_stack_f64_mid(const int* __restrict__ stack, int nstack,
                                                          const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                          double* __restrict__ c_data, int m, int n, int k, int group) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KSL = 8, TM = (RBX + 1) / 2, TN = (CBX + 1) / 2, PB = KSL + 4, KL = KSL / 2, CPR = 64 / KL;
  constexpr int ABYTES = mid_a_bytes(TM, KSL);
  constexpr int RA = (ABYTES + 1023) / 1024, RB_ = BT ? (mid_a_bytes(TN, KSL) + 1023) / 1024 : (8 * TN + CPR - 1) / CPR;
  const int lane = threadIdx.x;
  const int first = blockIdx.x * group;
  if (first >= nstack) return;
  const int last = min(first + group, nstack);
  const int bk = 2 * (lane & (KL - 1)), bc = lane / KL;
  u32x4 ga[RA], gb[RB_];
  int k0_cur = 0;
  auto issue = [&](int s, int k0) __attribute__((always_inline)) {
    const int ao = __builtin_amdgcn_readfirstlane(stack[3 * s]) - 1, bo = __builtin_amdgcn_readfirstlane(stack[3 * s + 1]) - 1;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + ao), 0, m * k * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + bo), 0, k * n * 8, 0x00020000);
    const int abase = lane * 16 + k0 * m * 8;   // (the whole offset in the bounds-checked operand: the k tail must arrive as zeros)
#pragma unroll
    for (int r = 0; r < RA; ++r) ga[r] = __builtin_amdgcn_raw_buffer_load_b128(rsa, abase + r * 1024, 0, 0);
    if constexpr (BT) {
      const int bbase = lane * 16 + k0 * n * 8;
#pragma unroll
      for (int r = 0; r < RB_; ++r) gb[r] = __builtin_amdgcn_raw_buffer_load_b128(rsb, bbase + r * 1024, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < RB_; ++r) {
        const int col = CPR * r + bc;
        const int off = col < n ? (col * k + k0 + bk) * 8 : 0x7ffffff0;
        gb[r] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off, 0, 0);
      }
    }
    k0_cur = k0;
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < RA; ++r) *reinterpret_cast<u32x4*>(smem + r * 1024 + lane * 16) = ga[r];
    DBCSR_AMD_LDS_ORDER();
    if constexpr (BT) {
#pragma unroll
      for (int r = 0; r < RB_; ++r) *reinterpret_cast<u32x4*>(smem + ABYTES + r * 1024 + lane * 16) = gb[r];
    } else {
      const bool k0ok = k0_cur + bk < k, k1ok = k0_cur + bk + 1 < k;
#pragma unroll
      for (int r = 0; r < RB_; ++r) {
        u32x4 v = gb[r];
        if (!k0ok) v[0] = 0u, v[1] = 0u;
        if (!k1ok) v[2] = 0u, v[3] = 0u;
        if (CPR * r + bc < 8 * TN) *reinterpret_cast<u32x4*>(smem + ABYTES + ((CPR * r + bc) * PB + bk) * 8) = v;
      }
    }
  };
  typedef BigSub<RBX, CBX> Sub;
  Sub S;
  S.init(0, 0, m, n, lane, ABYTES / 8, BT ? 1 : PB, BT ? n : 1);
  const int own_r = (m + 3) >> 2, own_c = (n + 3) >> 2;
  auto flush = [&](int co) __attribute__((always_inline)) {
    double* C = c_data + (co - 1);
    S.drain(lane, own_r, own_c, [&](int row, int col, double sum) {
      if (row < m && col < n) unsafeAtomicAdd(C + row + (size_t)m * col, sum);
    });
  };
  const double* la = reinterpret_cast<const double*>(smem);
  const int astep = 4 * m, bstep = BT ? 4 * n : 4;
  int s = first, k0 = 0;
  int cur_c = __builtin_amdgcn_readfirstlane(stack[3 * first + 2]);
  issue(first, 0);
  while (s < last) {
    if (k0 == 0) {   // a new entry: the end of a run of equal C offsets?
      const int co = __builtin_amdgcn_readfirstlane(stack[3 * s + 2]);
      if (co != cur_c) {
        flush(cur_c);
        cur_c = co;
      }
    }
    const int rem = (k - k0 + 3) >> 2;
    const int nst = rem < KSL / 4 ? rem : KSL / 4;
    stage();
    int s2 = s, k2 = k0 + KSL;
    if (k2 >= k) s2 = s + 1, k2 = 0;
    if (s2 < last) issue(s2, k2);
    // (one operand set: a step's fragments are fetched right before its MFMAs -- with two sets the 9 x 10 shapes do not fit three waves per SIMD here)
    double av[Sub::NA > 0 ? Sub::NA : 1], bv[Sub::NB > 0 ? Sub::NB : 1];
#pragma unroll
    for (int st_ = 0; st_ < KSL / 4; ++st_) {
      if (st_ < nst) {
        S.fetch(la, astep * st_, bstep * st_, av, bv);
        S.mma(av, bv);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    s = s2;
    k0 = k2;
  }
  flush(cur_c);
}

template <int RBX, int CBX>
static int launch_f64_mid(bool bt, hipStream_t st, const int* stack, int nstack, const double* a, const double* b, double* c, int m, int n, int k) {
  constexpr int TM = (RBX + 1) / 2, TN = (CBX + 1) / 2;
  const int group = 8;   // entries per wave (as the workgroup kernel: runs of equal C offsets of the host's stacks are about that long)
  const dim3 grid((unsigned)((nstack + group - 1) / group));
  const size_t a_b = (size_t)mid_a_bytes(TM, 8), lds_t = a_b + (size_t)((mid_a_bytes(TN, 8) + 1023) / 1024) * 1024,
               lds_s = std::max((a_b + 1023) / 1024 * 1024, a_b + (size_t)mid_b_bytes(TN, 8));
  if (bt)
    hipLaunchKernelGGL((smm_stack_f64_mid<RBX, CBX, true>), grid, dim3(64), std::max(lds_t, (a_b + 1023) / 1024 * 1024), st, stack, nstack, a, b, c, m, n, k, group);
  else
    hipLaunchKernelGGL((smm_stack_f64_mid<RBX, CBX, false>), grid, dim3(64), lds_s, st, stack, nstack, a, b, c, m, n, k, group);
  return dbcsr_amd::check(hipGetLastError(), "smm_stack_f64_mid launch", __FILE__, __LINE__);
}

// 1: not a shape of the one-wave kernel
static int process_stack_f64_mid(const int* dev_stack, int nstack, const double* a, const double* b, double* c, int m, int n, int k, bool bt, hipStream_t st) {
  const int rb = (m + 3) / 4, cb = (n + 3) / 4;
  switch (rb * 16 + cb) {
#define DBCSR_MID_STACK(A_, B_) \
  case A_ * 16 + B_: return launch_f64_mid<A_, B_>(bt, st, dev_stack, nstack, a, b, c, m, n, k);
    DBCSR_MID_STACK(9, 9) DBCSR_MID_STACK(9, 10) DBCSR_MID_STACK(10, 9) DBCSR_MID_STACK(10, 10)
#undef DBCSR_MID_STACK
    default: return 1;
  }
}

static int process_stack_f64_big(const int* dev_stack, int nstack, const double* a, const double* b, double* c, int m, int n, int k, bool bt,
                                 hipStream_t st) {
  const int tm = std::max(2, ((m + 7) / 8 + 1) / 2), tn = std::max(2, ((n + 7) / 8 + 1) / 2);
  switch (tm * 8 + tn) {
#define DBCSR_BIG_STACK(A_, B_) \
  case A_ * 8 + B_: return launch_f64_big<A_, B_>(bt, st, dev_stack, nstack, a, b, c, m, n, k);
    DBCSR_BIG_STACK(2, 2) DBCSR_BIG_STACK(2, 3) DBCSR_BIG_STACK(2, 4) DBCSR_BIG_STACK(2, 5)
    DBCSR_BIG_STACK(3, 2) DBCSR_BIG_STACK(3, 3) DBCSR_BIG_STACK(3, 4) DBCSR_BIG_STACK(3, 5)
    DBCSR_BIG_STACK(4, 2) DBCSR_BIG_STACK(4, 3) DBCSR_BIG_STACK(4, 4) DBCSR_BIG_STACK(4, 5)
    DBCSR_BIG_STACK(5, 2) DBCSR_BIG_STACK(5, 3) DBCSR_BIG_STACK(5, 4) DBCSR_BIG_STACK(5, 5)
#undef DBCSR_BIG_STACK
    default: return -1;
  }
}

// Homogeneous stacks of blocks up to 32 x 32 x 32 (round 6): the exact-size kernel of smm_exact.h, compiled the first time a stack of the
// triplet arrives -- as the reference does (libsmm_acc.cpp:90-195, 281-321).  Short stacks of a triplet not compiled yet do not pay for a
// compilation (~0.5 s); they -- and every stack when hiprtc is missing or DBCSR_AMD_SMM_EXACT=0 -- run the run-time-size kernels below.
// DBCSR_AMD_SMM_EXACT=1: every stack (tests).  Returns 1 when the stack was not taken.
static int process_stack_f64_exact(const int* dev_stack, int nstack, const double* a, const double* b, double* c, int m, int n, int k, bool bt,
                                   hipStream_t st) {
  static const int mode = getenv("DBCSR_AMD_SMM_EXACT") ? atoi(getenv("DBCSR_AMD_SMM_EXACT")) : -1;   // -1: automatic
  if (mode == 0) return 1;
  // blocks below 8^3: a stack is one launch of ~13 us whatever the kernel, and sixteen host threads launching through hipModuleLaunchKernel were
  // measured slower than through the ahead-of-time kernels (4^3: 0.17 against 0.13 ms per stack and thread, profiles/r06_acc_abi_threads.txt)
  if (mode < 0 && (int64_t)m * n * k < 512) return 1;
  struct Hit {
    int m = 0, n = 0, k = 0, bt = 0, dev = -1;
    StackKernel sk;
  };
  thread_local Hit last[4];   // the triplets this host thread met last (a host thread works through stacks of a few triplets in turn)
  thread_local int next = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1;
  const StackKernel* sk = nullptr;
  for (const Hit& h : last)
    if (h.dev == dev && h.m == m && h.n == n && h.k == k && h.bt == (int)bt) sk = &h.sk;
  if (!sk) {
    // a short stack of a triplet this thread has not met: the kernel another thread compiled, if any -- never a compilation of its own
    Hit& h = last[next];
    if (jit_stack_kernel(m, n, k, bt, &h.sk, !(mode < 0 && nstack < 256)) != 0) {
      h.dev = -1;
      return 1;
    }
    h.m = m, h.n = n, h.k = k, h.bt = (int)bt, h.dev = dev;
    next = (next + 1) & 3;
    sk = &h.sk;
  }
  const int group = stack_group(m, n, k);
  const int nwaves = (nstack + group - 1) / group;
  int group_arg = group;
  const int* stack_arg = dev_stack;
  void* args[] = {&stack_arg, &nstack, &a, &b, &c, &group_arg};
  note_kernel("smm_stack_f64_exact<%d,%d,%d%s>", m, n, k, bt);
  const hipError_t e = hipModuleLaunchKernel(sk->fn, (unsigned)((nwaves + 3) / 4), 1, 1, 256, 1, 1, (unsigned)(4 * sk->wave_lds), st, args, nullptr);
  return dbcsr_amd::check(e, "smm_stack_f64_exact launch", __FILE__, __LINE__);
}

int process_stack_f64(const int* dev_stack, int nstack, const double* a, const double* b, double* c, int m, int n, int k, bool bt,
                      hipStream_t st) {
  if (nstack <= 0) return 0;
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  if (m <= 32 && n <= 32 && k <= 32) {
    const int rc = process_stack_f64_exact(dev_stack, nstack, a, b, c, m, n, k, bt, st);
    if (rc <= 0) return rc;
  }
  // blocks of 33 ... 80 (or an inner dimension above 32): a workgroup per group of entries, operand slabs shared through LDS
  static const bool big_off = getenv("DBCSR_AMD_SMM_BIG") != nullptr && atoi(getenv("DBCSR_AMD_SMM_BIG")) == 0;
  // (a long inner dimension alone takes this path only when the C block has work for the four waves of a workgroup -- at least 2 x 2 tiles of
  // 8 x 8: for 5 x 5 x 64 three of them would stage slabs and wait at barriers for nothing; those stay with the wave-per-entry kernels below)
  const bool wide = m > 32 || n > 32 || (k > 32 && ((m + 7) / 8) * ((n + 7) / 8) >= 4);
  if (!big_off && wide && m <= 80 && n <= 80 && (int64_t)m * k * 8 < (1ll << 30) && (int64_t)n * k * 8 < (1ll << 30)) {
    // blocks of 33 ... 40 in both dimensions: one wave per group of entries, the block in units of 4 x 4 (DBCSR_AMD_SMM_MID=0: the workgroup kernel)
    static const bool mid_off = getenv("DBCSR_AMD_SMM_MID") != nullptr && atoi(getenv("DBCSR_AMD_SMM_MID")) == 0;
    if (!mid_off && m > 32 && n > 32 && m <= 40 && n <= 40) {
      note_kernel("smm_stack_f64_mid(%d,%d,%d%s)", m, n, k, bt);
      const int rc = process_stack_f64_mid(dev_stack, nstack, a, b, c, m, n, k, bt, st);
      if (rc <= 0) return rc;
    }
    note_kernel("smm_stack_f64_big(%d,%d,%d%s)", m, n, k, bt);
    return process_stack_f64_big(dev_stack, nstack, a, b, c, m, n, k, bt, st);
  }
  note_kernel(m <= 32 && n <= 32 && k <= 32 ? "smm_stack_f64_lds(%d,%d,%d%s)" : "smm_stack_f64(%d,%d,%d%s)", m, n, k, bt);
  // C tile per wave: up to 32 x 32; larger blocks are tiled over grid.y/z
  const int MA = m >= 32 ? 4 : (m + 7) / 8, NC = n >= 32 ? 4 : (n + 7) / 8;
  const int tiles_r = (m + 8 * MA - 1) / (8 * MA), tiles_c = (n + 8 * NC - 1) / (8 * NC);
  const int group = stack_group(m, n, k);
  const int nwaves = (nstack + group - 1) / group;
  dim3 grid((nwaves + 3) / 4, tiles_r, tiles_c);
  switch (MA) {
    case 1: return launch_f64_nc<1>(NC, bt, grid, st, dev_stack, nstack, a, b, c, m, n, k, group);
    case 2: return launch_f64_nc<2>(NC, bt, grid, st, dev_stack, nstack, a, b, c, m, n, k, group);
    case 3: return launch_f64_nc<3>(NC, bt, grid, st, dev_stack, nstack, a, b, c, m, n, k, group);
    default: return launch_f64_nc<4>(NC, bt, grid, st, dev_stack, nstack, a, b, c, m, n, k, group);
  }
}

int process_stack_f32(const int* dev_stack, int nstack, const float* a, const float* b, float* c, int m, int n, int k, bool bt,
                      hipStream_t st) {
  if (nstack <= 0) return 0;
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  const int group = stack_group(m, n, k);
  const int nwaves = (nstack + group - 1) / group;
  dim3 grid((nwaves + 3) / 4, (m + 31) / 32, (n + 31) / 32);
  if (!bt && m <= 32 && n <= 32 && k <= 32) {
    hipLaunchKernelGGL(smm_stack_f32_lds, grid, dim3(256), 0, st, dev_stack, nstack, a, b, c, m, n, k, group);
    return dbcsr_amd::check(hipGetLastError(), "smm_stack_f32_lds launch", __FILE__, __LINE__);
  }
  if (bt)
    hipLaunchKernelGGL((smm_stack_f32<true>), grid, dim3(256), 0, st, dev_stack, nstack, a, b, c, m, n, k, group);
  else
    hipLaunchKernelGGL((smm_stack_f32<false>), grid, dim3(256), 0, st, dev_stack, nstack, a, b, c, m, n, k, group);
  return dbcsr_amd::check(hipGetLastError(), "smm_stack_f32 launch", __FILE__, __LINE__);
}

// ---- inhomogeneous stacks ------------------------------------------------------------------------------------------------
// A stack whose entries have different (m, n, k) -- the products with the tail blocks, or a fourth block size: everything the
// host's three most common sizes per dimension do not cover (dbcsr_mm_csr.F:236-248) -- is refused by the reference's library
// (libsmm_acc.cpp:324-339 returns -1: the host multiplies it on the CPU, and aborts in G2G mode, dbcsr_mm_accdrv.F:534).  Here
// it runs on the device: the sizes are in the host's own 7-integer records (m, n, k, a, b, c, c_blk; dbcsr_mm_types.F:24-37),
// which are copied to the device for the call; one wavefront per entry, C tiles of 32 x 32, fp64 atomics into C.
__global__ void __launch_bounds__(256) smm_stack_f64_mixed(const int* __restrict__ params, int nstack, const double* __restrict__ a_data,
                                                           const double* __restrict__ b_data, double* __restrict__ c_data, int max_dim) {
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (s >= nstack) return;
  const int* p = params + 7 * (size_t)s;
  const int m = __builtin_amdgcn_readfirstlane(p[0]), n = __builtin_amdgcn_readfirstlane(p[1]), k = __builtin_amdgcn_readfirstlane(p[2]);
  const int ao = __builtin_amdgcn_readfirstlane(p[3]), bo = __builtin_amdgcn_readfirstlane(p[4]), co = __builtin_amdgcn_readfirstlane(p[5]);
  if (m <= 0 || n <= 0 || k <= 0) return;
  const bool bt = k <= max_dim && n <= max_dim;  // the host transposed this B block (libsmm_acc_transpose) iff both of its dims are small
  const LaneMap L(lane);
  const double* A = a_data + (ao - 1);
  const double* B = b_data + (bo - 1);
  double* C = c_data + (co - 1);
  for (int row0 = 0; row0 < m; row0 += 32)
    for (int col0 = 0; col0 < n; col0 += 32) {
      double acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.0;
      if (bt)
        block_product_f64<4, 4, true>(acc, A, B, m, n, k, L, row0, col0);
      else
        block_product_f64<4, 4, false>(acc, A, B, m, n, k, L, row0, col0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int row = row0 + 8 * a + L.rowd, col = col0 + 8 * c + L.coll;
          if (row < m && col < n) unsafeAtomicAdd(C + row + (size_t)m * col, acc[a][c]);
        }
    }
}

// device copy of the host records, one buffer per stream (calls on one stream are ordered; the host calls from several OpenMP
// threads with their own streams), and a PINNED staging buffer of the library's own next to it: the host reuses its stack right
// after the call returns, and only a copy from pageable memory has read its source by then -- a host that pins or registers its
// stack buffers (the reference pins stackbuf%hostmem; a CP2K build may hipHostRegister) would get a truly asynchronous copy and
// silently wrong products.  So the records are copied on the host into the staging buffer first; an event says when the device
// has taken them over from there.
struct MixedScratch {
  std::mutex mu;
  struct Slot {
    int* dev = nullptr;
    int* pinned = nullptr;
    size_t cap = 0;  // ints
    hipEvent_t taken = nullptr;
    bool pending = false;
  };
  std::map<hipStream_t, Slot> buf;
  Slot* get(hipStream_t st, size_t nints) {
    std::lock_guard<std::mutex> lk(mu);
    Slot& e = buf[st];
    if (!e.taken && hipEventCreateWithFlags(&e.taken, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (e.cap < nints) {
      if (e.pending) (void)hipEventSynchronize(e.taken);
      e.pending = false;
      if (e.dev) (void)hipFree(e.dev);  // (hipFree waits for the device: the old buffer is no longer in use)
      if (e.pinned) (void)hipHostFree(e.pinned);
      e.dev = e.pinned = nullptr;
      e.cap = 0;
      const size_t want = nints + nints / 4 + 1024;
      if (hipMalloc(reinterpret_cast<void**>(&e.dev), want * sizeof(int)) != hipSuccess) return nullptr;
      if (hipHostMalloc(reinterpret_cast<void**>(&e.pinned), want * sizeof(int), hipHostMallocDefault) != hipSuccess) return nullptr;
      e.cap = want;
    }
    return &e;  // (std::map nodes do not move: the pointer stays valid; one stream is used by one host thread at a time)
  }
};
static MixedScratch g_mixed;

int process_stack_f64_mixed(const int* host_params, int nstack, const double* a, const double* b, double* c, int max_dim, hipStream_t st) {
  if (nstack <= 0) return 0;
  if (!host_params) return -1;
  MixedScratch::Slot* e = g_mixed.get(st, (size_t)7 * nstack);
  if (!e) return -1;
  if (e->pending) ACC_CHECK(hipEventSynchronize(e->taken));  // the previous stack of this stream has left the staging buffer
  memcpy(e->pinned, host_params, sizeof(int) * 7 * (size_t)nstack);
  ACC_CHECK(hipMemcpyAsync(e->dev, e->pinned, sizeof(int) * 7 * (size_t)nstack, hipMemcpyHostToDevice, st));
  ACC_CHECK(hipEventRecord(e->taken, st));
  e->pending = true;
  hipLaunchKernelGGL(smm_stack_f64_mixed, dim3((nstack + 3) / 4), dim3(256), 0, st, e->dev, nstack, a, b, c, max_dim);
  return dbcsr_amd::check(hipGetLastError(), "smm_stack_f64_mixed launch", __FILE__, __LINE__);
}

}  // namespace dbcsr_amd

using namespace dbcsr_amd;

extern "C" {

int libsmm_acc_init(void) {
  // libsmm_acc_init.cpp:60-73 checks the wavefront width; gfx950 is 64-wide.
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;  // no device: nothing to initialise (the host decides what to do)
  }
  hipDeviceProp_t prop;
  ACC_CHECK(hipGetDeviceProperties(&prop, dev));
  if (prop.warpSize != 64) {
    fprintf(stderr, "dbcsr_acc_amd: wavefront size %d != 64 (this library targets gfx950 only)\n", prop.warpSize);
    return -1;
  }
  return 0;
}

int libsmm_acc_finalize(void) { return 0; }

// All entry points are re-entrant (no shared mutable state): every OpenMP
// thread of the host may call with its own streams (core/dbcsr_lib.F:248-262).
c_dbcsr_acc_bool_t libsmm_acc_is_thread_safe(void) { return 1; }

int libsmm_acc_gpu_warp_size(void) { return 64; }

const char* dbcsr_amd_smm_last_kernel(void) { return g_last_smm_kernel; }

int libsmm_acc_transpose(const int* dev_trs_stack, int offset, int stack_size, void* dev_data, libsmm_acc_data_t datatype, int m,
                         int n, int max_kernel_dim, void* stream) {
  if (datatype != dbcsr_type_real_8) return 0;               // transpose not needed (libsmm_acc.cpp:484)
  if (m > max_kernel_dim || n > max_kernel_dim) return 0;    // (libsmm_acc.cpp:485)
  if (stack_size <= 0 || m <= 0 || n <= 0) return 0;
  const size_t lds = sizeof(double) * (size_t)m * n;
  hipLaunchKernelGGL(transpose_blocks_f64, dim3(stack_size), dim3(m * n >= 256 ? 256 : (m * n > 64 ? 128 : 64)), lds,
                     stream_of(stream), dev_trs_stack + offset, static_cast<double*>(dev_data), m, n);
  return dbcsr_amd::check(hipGetLastError(), "transpose_blocks_f64 launch", __FILE__, __LINE__);
}

int libsmm_acc_process(const int* host_param_stack, const int* dev_param_stack, int stack_size, libsmm_acc_data_t datatype,
                       const void* dev_a_data, const void* dev_b_data, void* dev_c_data, int m_max, int n_max, int k_max,
                       int max_kernel_dim, c_dbcsr_acc_bool_t def_mnk, void* stack_stream, void* c_stream) {
  (void)c_stream;
  if (def_mnk != 1) {
    // inhomogeneous stack: on the device from the host's own records for fp64 (see smm_stack_f64_mixed), host path otherwise
    static const bool mixed_on_host = getenv("DBCSR_AMD_SMM_MIXED_ON_HOST") != nullptr;  // (read once: this is the call path of every OpenMP thread)
    if (datatype != dbcsr_type_real_8 || !host_param_stack || mixed_on_host) return -1;
    return process_stack_f64_mixed(host_param_stack, stack_size, static_cast<const double*>(dev_a_data), static_cast<const double*>(dev_b_data),
                                   static_cast<double*>(dev_c_data), max_kernel_dim, stream_of(stack_stream));
  }
  if (datatype == dbcsr_type_real_8) {
    // B was transposed by libsmm_acc_transpose iff its own dims (k x n) are both <= max_kernel_dim
    const bool bt = (k_max <= max_kernel_dim && n_max <= max_kernel_dim);
    return process_stack_f64(dev_param_stack, stack_size, static_cast<const double*>(dev_a_data),
                             static_cast<const double*>(dev_b_data), static_cast<double*>(dev_c_data), m_max, n_max, k_max, bt,
                             stream_of(stack_stream));
  }
  if (datatype == dbcsr_type_real_4) {
    return process_stack_f32(dev_param_stack, stack_size, static_cast<const float*>(dev_a_data),
                             static_cast<const float*>(dev_b_data), static_cast<float*>(dev_c_data), m_max, n_max, k_max, false,
                             stream_of(stack_stream));
  }
  return -10;  // complex types: host path
}

int c_calculate_norms(const double* mat, int nblks, const int* offsets, const int* nelems, float* norms, void* stream_ptr) {
  if (nblks <= 0) return 0;
  hipLaunchKernelGGL(block_norms_f64, dim3((nblks + 3) / 4), dim3(256), 0, stream_of(stream_ptr), mat, nblks, offsets, nelems,
                     norms);
  return dbcsr_amd::check(hipGetLastError(), "block_norms_f64 launch", __FILE__, __LINE__);
}

}  // extern "C"
