// mm_engine.hip -- device-resident local multiply (include/dbcsr_amd_mm.h).
//
// Symbolic phase (integer, HBM/L2-bound): block-level bitmaps.
//   Bbm[k][w]  : bit j set iff B(k,j) present                 (bitmap_from_index)
//   Cbm[i][w]  = Cin_bm[i][w] | OR_{k in A-row(i)} Bbm[k][w]  (c_bitmap)
//   row prefix popcounts give, without any hashing, the sorted column index of
//   C (what dbcsr_finalize produces, work/dbcsr_work_operations.F:749+) and the
//   rank of any block inside its row (row_prefix).
//   For every C block the list of products (a_off, b_off, k) is emitted in
//   ascending k: deterministic, no atomics (count_products / fill_products).
// This restates WHAT dbcsr_mm_csr_multiply_low computes (mm/dbcsr_mm_csr.F:
// 257-357: which C blocks exist, which (A,B) pairs feed each) with a data-
// parallel algorithm instead of its per-thread hash tables and 30000-entry
// parameter stacks.
//
// Numeric phase (fp64/fp32 MFMA): one wavefront per C block, all products of
// the block accumulated in registers, C written exactly once (no atomics, no
// zero-fill pass, bitwise reproducible).  Kernels, chosen per launch by the host code at the end of this file:
//   mm_numeric_f64_hot<M,N,K> / mm_numeric_f32_hot<M,N,K>  exact-size kernels, one (m, n, k) dominates (cubes 9..32)
//   mm_numeric_f64_tiny                                    C blocks of at most 4 x 4: four C blocks per wave
//   mm_numeric_f64_lds<MAXT> / mm_numeric_f64_pipe<MAXT>   any sizes up to 32 (pipe: mixed sizes, few products per block)
//   mm_numeric_f32_lds                                     fp32, any sizes up to 32
//   mm_numeric_f64 / mm_numeric_f32                        blocks above 32 (32 x 32 tiles, fragments from global memory)
// Around them: transpose, checksum, synthetic fill, norm filter, crop / window scale (submatrix limits).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <vector>

#include "../../include/dbcsr_amd_mm.h"
#include "common.h"
#include "smm_core.h"
#include "mm_types.h"
#include "mm_jit.h"

#include "mm_workspace.h"
#include "mm_symbolic.h"
#include "mm_numeric_f64.h"
#include "mm_numeric_f64_big.h"
#include "mm_mid.h"   // the one-wave slab kernels of the blocks of 25 ... 40 (mm_numeric_f64_mid.h, mm_mid.hip)
#include "mm_numeric_f32.h"
#include "mm_aux.h"
// The library comes in two builds (Makefile): the SHIPPING one holds what a multiply can run by itself -- the kernels listed above, their
// symbolic phases, plan reuse -- and the LAB one (-DDBCSR_AMD_EXPERIMENTS, libdbcsr_acc_amd_lab.so) adds every dataflow and variant that
// was built, made parity-green and measured but does not win: the LDS-DMA ring kernels (mm_dma.h), XCD-wide C tiles (mm_tile.*), CU-wide
// C tiles with B shared in LDS (mm_band.*), the persistent form and the ablation / keep-alive variants of the exact-size kernel, the
// G-block bodies and stream spreading of the class kernels, the occupancy and row-group knobs.  Their switches exist in the lab build only.
#ifdef DBCSR_AMD_EXPERIMENTS
#include "mm_lab_api.h"
#include "mm_group.h"
#include "mm_group64.h"
#include "mm_dma.h"
#include "mm_tile_index.h"
#include "mm_band_index.h"
#endif
namespace dbcsr_amd {

// ---- plan reuse ------------------------------------------------------------------------------------------------------
// A multiply whose operands have the SAME index arrays (patterns, block sizes, block offsets) as the previous multiply of the
// engine -- every SCF step of a CP2K run, every repetition of the performance driver -- needs no new symbolic phase: the engine
// keeps device copies of the last call's index arrays and compares the incoming ones word by word (one small kernel, one flag).
struct PlanSegs {
  const int32_t* a[12];
  const int32_t* b[12];
  long long n[12];  // 32-bit words per segment
  int nseg;
};
__global__ void __launch_bounds__(256) plan_compare(PlanSegs S, int* __restrict__ differs) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  bool bad = false;
  for (int g = 0; g < S.nseg; ++g)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < S.n[g]; i += stride) bad |= S.a[g][i] != S.b[g][i];
  if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(differs, 1);
}

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

struct Engine {
  DevBuf<uint32_t> b_bm, c_bm, cin_bm;
  DevBuf<int> b_pre, c_pre, cin_pre, row_nnz, prod_cnt, blk_nze, tmp_i32;
  DevBuf<int64_t> prod_start, c_blk_p_ws, partial, off_a, off_b;
  DevBuf<Entry> entries;
  DevBuf<Desc> descs;
  DevBuf<double> row_sums, norms64;
  DevBuf<float> a_norms, b_norms;
  DevBuf<int> keep;
  FilterArgs filter = {nullptr, nullptr, 0.0f};
  int64_t flt_nblks = 0;
  DevBuf<int> order, order_cnt;
  const void* norms_data = nullptr;  // norms64[] holds the block norms of the matrix with this data pointer (left by the numeric kernel)
  int64_t norms_nblks = 0;
  int canonical_c = 0;  // dbcsr_amd_mm_set_canonical_product: the product matrix has symmetry, its index is in canonical form
  // fp32: a wave owns R C blocks of one block column and shares B among them (mm_group.h).  DBCSR_AMD_MM_F32_GROUP = 2 / 3 / 4: that R
  // whenever the kernel applies; -1: R = 4 when C blocks have at least 16 products on average; 0 / unset: off -- measured (gpurun_out/r05_s04:
  // 32768^2 at 20 % fill 32.1 ms against 28.6 for one wave per block, config 5 2125 against 1836 ms) it trades B blocks over the fabric for
  // A rows that no longer fit the XCD's L2 and for occupancy (3 waves per SIMD instead of 5), and loses
  int f32_group = 0, group_R = 0;
  bool group_built = false, b_monotone = false;
  DevBuf<int> groups, group_flag;
  // fp64 (round 6, mm_group64.h): DBCSR_AMD_MM_F64_GROUP = 2 ... 6: a wave owns that many C blocks of one block column whenever the kernel
  // applies; 0 / unset: off.  DBCSR_AMD_MM_GROUP_PANEL_MB: target size of a B column panel of the group launch (0: panel_bytes)
  int f64_group = 0;
  int64_t group_panel_bytes = 0;
  DevBuf<int> group_cnt;
#ifdef DBCSR_AMD_EXPERIMENTS
  DevBuf<GWork> group_work;
  DevBuf<GEntry> group_entries;
#endif
  DevBuf<int64_t> group_start;
  int use_mid = 1;     // DBCSR_AMD_MM_MID=0: blocks of 33 ... 40 through the workgroup kernel mm_numeric_f64_big instead of the one-wave kernel mm_numeric_f64_mid
  int use_big = 1;     // DBCSR_AMD_MM_BIG=0: blocks above 32 through the one-wave-per-block kernel of rounds 1-4 (mm_numeric_f64) instead of mm_numeric_f64_big
  int f32_direct = 1;  // (2: + the slim-LDS launch when every C block has the dominant size -- more waves per CU, measured 0-4 % slower: the
                       // kernel is fabric-bound, gpurun_out/r05_s17 --, 1: never slim) DBCSR_AMD_MM_F32_DIRECT=0: the fp32 exact-size kernel that stages both operands in LDS (rounds 1-4) instead of the direct form
  int wg_waves = 0;   // DBCSR_AMD_MM_WG_WAVES = 1 | 2 | 4: waves per workgroup of the one-wave-per-C-block kernels (0: by list length).  A workgroup's LDS is
                      // released when its LAST wave ends, so with product lists of uneven length fewer waves per workgroup keep
                      // more of the CU's wave slots busy (config 3: kernel 8.93 / 8.09 / 7.51 ms for 4 / 2 / 1, config 2: 23.6 / 22.7 /
                      // 22.6, config 4: 30.1 / 28.8 / 28.4 on the same box, profiles/r02_wg_waves_bench_lines.txt)
  DevBuf<Work> work;  // launch-order records of the exact-size fp64 kernels (DBCSR_AMD_MM_WORK=0: the class kernels read order[] -> descs[] -> entries[] instead)
  int use_work = 1;
  DevBuf<int64_t> order_base;
  int64_t order_len = 0;
  Window crop_win = {0, 0, 0, 0};       // window of the last dbcsr_amd_bcsr_crop_count
  bool crop_pending = false;
  int hot_m = 0, hot_n = 0, hot_k = 0;  // dominant block sizes of the last symbolic phase (0: none)
  int use_tiny = 1;                     // DBCSR_AMD_MM_TINY=0: no packed kernel for blocks of at most 4 x 4
  int use_hot = 1;                      // DBCSR_AMD_MM_HOT=0: never use the exact-size kernels
  int lds_pad = 0;                      // DBCSR_AMD_MM_LDS_PAD: extra LDS bytes per workgroup (occupancy experiments)
  int64_t panel_bytes = 256ll << 20;  // DBCSR_AMD_MM_PANEL_MB: target size of a B column panel (config 2, round 3: 160 / 200 / 256 / 320 / 400 MB ->
                                      // 18.97 / 18.76 / 18.63 / 18.95 / 19.04 ms, profiles/r03_panel_wgwaves_sweep.txt)
  int row_group = 0;                 // DBCSR_AMD_MM_ROW_GROUP: rows walked together per XCD (0 = automatic)
  DevBuf<unsigned long long> dev_scalars, stat_table;
  int64_t* host_scalars = nullptr;  // pinned: [0]=c_nblks [1]=c_nze [2]=nproducts [3]=flop
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // around fill_products and the numeric kernel
  bool timed = false;
  // state carried from symbolic to numeric
  int nbr = 0, W = 0;
  int64_t c_nblks = 0, nproducts = 0;
  bool have_cin = false, retain = false, valid = false;
  int max_m = 0, max_k = 0, max_n = 0, min_m = 0, min_k = 0, min_n = 0;
  bool rows_kernels = false;  // product-driven symbolic kernels (sparse C); DBCSR_AMD_MM_SYMBOLIC=rows forces, =grid / =word exclude
  int force_symbolic = 0;     // 0 automatic, 1 word, 2 grid, 3 rows
  bool grid_kernels = false, force_word_kernels = false;  // DBCSR_AMD_MM_SYMBOLIC=word forces the per-word symbolic kernels
  int dbg = 0;      // DBCSR_AMD_MM_DBG: ablation switches of the LDS kernel (profiling only; the exact-size kernel honours them in its VAR = 1 build)
  // XCD-wide C tiles in registers (mm_tile.h): DBCSR_AMD_MM_TILE = 0 never, 1 automatic, 2 whenever the sizes allow;
  // DBCSR_AMD_MM_TILE_WINDOW = k window of the team (inner blocks; 0: no throttle); DBCSR_AMD_MM_TILE_RDV = 1: unpaired fragment reads
  int tile_shape = 0;  // DBCSR_AMD_MM_TILE_SHAPE: 0 = 3 x 3 C blocks per wave, two waves per SIMD; 1 = 4 x 3, one wave per SIMD, four-slot ring (mm_tile.h)
  int use_tile = 0, tile_window = 256, tile_rdv = 0, tile_pub = 1, tile_prefetch = 0, tile_knobs = 0;  // DBCSR_AMD_MM_TILE_PUB: progress stores written through (0) / left in L2 (1)
  int hot_cnt_m = 0, hot_cnt_k = 0, hot_cnt_n = 0;  // block rows / inner blocks / block columns of the dominant size
  DevBuf<uint32_t> a_bm, bt_bm, tile_prog;
  DevBuf<unsigned long long> tile_times;
  DevBuf<int> a_pre, tile_rows, tile_cols, tile_cnt, tile_flags;
  DevBuf<int64_t> tile_start;
#ifdef DBCSR_AMD_EXPERIMENTS
  DevBuf<TileDesc> tdescs;
  DevBuf<TileEntry> tentries;
#endif
  // plan reuse (plan_compare): device copies of the index arrays the last symbolic phase saw, C's index as the numeric phase emitted it
  int use_plan = 1;  // DBCSR_AMD_MM_PLAN=0: every multiply runs its symbolic phase
  bool plan_saved = false, plan_hit = false, plan_numeric = false;
  int plan_dims[3] = {0, 0, 0}, plan_retain = 0, plan_canonical = 0, plan_datatype = 0;
  int64_t plan_nblks[3] = {0, 0, 0};
  // dbcsr_amd_mm_trust_plan: index arrays at the ADDRESSES the saved plan saw are taken as unchanged (no comparison on the device, no
  // synchronisation): for callers that own their operands' index and never write it in place
  bool plan_trusted = false;
  const void* plan_ptrs[12] = {nullptr};
  uint64_t plan_stamps[3] = {0, 0, 0};  // index_stamp of A, B, C_in when the plan was saved (0: unknown generation, never trusted)
  DevBuf<int32_t> plan_words, plan_c_col_i;
  DevBuf<int64_t> plan_c_blk_p;
  DevBuf<int> plan_flag;
  int* plan_host_flag = nullptr;  // pinned
  dbcsr_amd_mm_counts plan_counts = {0, 0, 0, 0};
  bool work_built = false, tile_built = false, band_built = false;
#ifdef DBCSR_AMD_EXPERIMENTS
  TileGeom tile_geom = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // CU-wide C tiles, B shared in an LDS ring (mm_band.h): DBCSR_AMD_MM_BAND = 0 never, 1 automatic, 2 whenever the sizes allow;
  // DBCSR_AMD_MM_BAND_DEPTH = slots of the ring (12 | 16 | 20 | 22); DBCSR_AMD_MM_BAND_BPOL = 1: B copies with the nt hint;
  // DBCSR_AMD_MM_BAND_KNOBS bit 0: where the waves' time goes (printed by dbcsr_amd_mm_band_stats)
  // DBCSR_AMD_MM_BAND_WINDOW = k window of an XCD's waves (inner blocks; 0: no throttle)
  // DBCSR_AMD_MM_BAND_SHAPE: 0 = 8 waves x (3 x 3 C blocks), 1 = 16 waves x (2 x 2)
  int use_band = 0, band_shape = 1, band_depth = 20, band_bpol = 0, band_knobs = 0, band_window = 384;
  int64_t band_nlist = 0, band_nrem = 0;
  DevBuf<unsigned> band_prog;
#ifdef DBCSR_AMD_EXPERIMENTS
  BandGeom band_geom = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  DevBuf<BandDesc> band_descs_buf;
  DevBuf<BandEntry> band_entries;
  DevBuf<BandRem> band_rem;
#endif
  DevBuf<int> band_cnt_list, band_cnt_b, band_cnt_rem, band_sub_cnt, band_flags;
  DevBuf<int64_t> band_list_off, band_seq_off, band_rem_start;
  DevBuf<unsigned long long> band_times;
  long long plan_hits = 0, plan_misses = 0;
  int hot_persistent = 0;  // DBCSR_AMD_MM_HOT_PERSISTENT=1: the 23^3 kernel as persistent waves with a work counter per XCD (mm_numeric_f64.h)
  unsigned hot_xcd_mask = 0xffu;  // DBCSR_AMD_MM_HOT_XCDS: XCDs the persistent form runs on (experiments: the others' C blocks are NOT computed)
  DevBuf<unsigned> hot_counters;
  int hot_variant = 0;  // DBCSR_AMD_MM_HOT_VARIANT: 2 = exact-size kernel with unpaired ds_read_b64 fragment reads (23^3 only)
  int use_pipe = -1, pipe_g = 8;  // multi-block pipelined kernel: -1 automatic (short product lists only), DBCSR_AMD_MM_KERNEL=pipe|lds1 forces; DBCSR_AMD_MM_PIPE_G = blocks per wave
  // (m, n) classes (mixed block sizes, see order_count_cls): DBCSR_AMD_MM_CLASSES = 0 never, 1 automatic, 2 always when the sizes allow
  int use_classes = 1;
  int class_g = 1;  // DBCSR_AMD_MM_CLASS_G: C blocks per wave in the class kernels (1, 2, 4, 8)
  bool cls_mode = false;
  // DBCSR_AMD_MM_CLASS_STREAMS: the class launches of one multiply touch disjoint C blocks; with n > 1 they are spread over n streams
  // (the caller's + n - 1 of the engine's, forked / joined with events) so that the tail of one launch overlaps the body of the next
  int class_streams = 1;
  hipStream_t side_stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t fork_ev = nullptr, join_ev[3] = {nullptr, nullptr, nullptr};
  int cls_m[3] = {0, 0, 0}, cls_n[3] = {0, 0, 0}, cls_k[3] = {0, 0, 0};
  int64_t cls_len[kNumClasses] = {0}, cls_off[kNumClasses] = {0};
  DevBuf<int> cls_hist;
  DevBuf<unsigned char> cls_row, cls_col;
  DevBuf<uint32_t> cls_col_bm;
  DevBuf<int64_t> cls_lens;
  int* cls_host_hist = nullptr;       // pinned: 3 x 33 size histograms
  int64_t* cls_host_lens = nullptr;   // pinned: 10 lengths, 10 offsets, total
  char last_kernel[96] = "";  // name of the numeric kernel of the last dbcsr_amd_mm_numeric (dbcsr_amd_mm_last_kernel)
  int dma_stages = 0;  // DBCSR_AMD_MM_KERNEL=dma2|dma3|dma4: LDS-DMA exact-size kernel with that many ring slots (0: off)
  int use_lds = 1;  // DBCSR_AMD_MM_KERNEL=direct selects the v1 kernel (A/B experiments)
  KPassMemo kpass_memo;  // dbcsr_amd_multiply's k-pass decision for the last stamped A operand (mm_api.hip)
};

KPassMemo* engine_kpass_memo(void* handle) { return handle ? &static_cast<Engine*>(handle)->kpass_memo : nullptr; }

// waves per block row for the kernels that stream whole blocks (norms, compaction): enough waves to keep the memory system busy
static inline int row_split(int64_t nbr, int64_t nblks) {
  if (nbr <= 0) return 1;
  const int64_t per_row = nblks / nbr;
  int64_t S = (65536 + nbr - 1) / nbr;
  if (S > per_row) S = per_row;
  return (int)std::max<int64_t>(1, std::min<int64_t>(S, 64));
}

template <typename TO>
static int exclusive_scan(Engine* E, const int* in, int64_t n, TO* out, int64_t* total_dev, bool write_total_at_n, hipStream_t st) {
  const int nb = (int)((n + kScanChunk - 1) / kScanChunk);
  if (E->partial.ensure((size_t)(nb > 0 ? nb : 1))) return -1;
  if (n <= 0) {
    if (total_dev) ACC_CHECK(hipMemsetAsync(total_dev, 0, sizeof(int64_t), st));
    if (write_total_at_n) ACC_CHECK(hipMemsetAsync(out, 0, sizeof(TO), st));
    return 0;
  }
  hipLaunchKernelGGL(scan_reduce, dim3(nb), dim3(kScanThreads), 0, st, in, n, E->partial.p);
  hipLaunchKernelGGL(scan_partials, dim3(1), dim3(kScanThreads), 0, st, E->partial.p, nb, total_dev);
  hipLaunchKernelGGL((scan_apply<TO>), dim3(nb), dim3(kScanThreads), 0, st, in, n, E->partial.p, out, write_total_at_n ? 1 : 0);
  return check(hipGetLastError(), "exclusive_scan", __FILE__, __LINE__);
}

static inline dim3 grid_for(int64_t nthreads) { return dim3((unsigned)((nthreads + 255) / 256)); }

#ifdef DBCSR_AMD_EXPERIMENTS
// fp32 group kernel: 0 = launched, 1 = does not apply here (the caller runs the one-wave-per-block kernel), < 0 = error
static int run_group_f32(Engine* E, int R, bool reuse, hipStream_t st, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
                         dbcsr_amd_bcsr* c_out, float alpha, float beta, int skip_empty) {
  const int S = E->hot_m, nbr = E->nbr, nbc = b->nblkcols;
  if (!(S == 16 || S == 24 || S == 32) || R < 2 || R > 4 || nbr <= 0 || nbc <= 0) return 1;
  const int ng = (nbr + R - 1) / R, ngx = (ng + 7) / 8;
  if ((int64_t)ngx * nbc >= (1ll << 30)) return 1;
  if (!(reuse && E->group_built && E->group_R == R)) {
    if (E->group_flag.ensure(4) || E->groups.ensure((size_t)ng * nbc * R + 1)) return -1;
    ACC_CHECK(hipMemsetAsync(E->group_flag.p, 0, sizeof(int), st));
    group_check_ascending(st, static_cast<const int64_t*>(b->blk_p), (int64_t)b->nblks, E->group_flag.p);
    int* hflag = reinterpret_cast<int*>(E->host_scalars + 12);
    ACC_CHECK(hipMemcpyAsync(hflag, E->group_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    ACC_CHECK(hipStreamSynchronize(st));
    E->b_monotone = *hflag == 0;
    E->group_R = R;
    E->group_built = true;
    if (E->b_monotone) group_build_table(st, c_out->row_p, c_out->col_i, E->descs.p, nbr, nbc, R, S, E->groups.p);
  }
  if (!E->b_monotone) return 1;
  GroupGeom G;
  G.nbc = nbc, G.ng = ng, G.ngx = ngx;
  const int64_t b_bytes = (int64_t)b->nblks * S * S * (int64_t)sizeof(float);
  int np = (int)std::min<int64_t>(std::max<int64_t>(1, (b_bytes + E->panel_bytes - 1) / E->panel_bytes), (int64_t)nbc);
  G.pw = (nbc + np - 1) / np;
  G.np = (nbc + G.pw - 1) / G.pw;
  const unsigned nwg = 8u * (unsigned)(((int64_t)ngx * nbc + 3) / 4);
  const float* ad = static_cast<const float*>(a->data);
  const float* bd = static_cast<const float*>(b->data);
  float* cd = static_cast<float*>(c_out->data);
  const float* cid = static_cast<const float*>(c_in->data);
  return group_f32_launch(S, R, nwg, st, E->descs.p, E->entries.p, ad, bd, cd, cid, alpha, beta, skip_empty, E->groups.p, G);
}

// fp64 group kernel (mm_group64.h): 0 = launched, 1 = does not apply here (the caller runs the one-wave-per-block kernel), < 0 = error
static int run_group_f64(Engine* E, int R, bool reuse, hipStream_t st, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
                         dbcsr_amd_bcsr* c_out, double alpha, double beta, int skip_empty) {
  const int S = E->hot_m, nbr = E->nbr, nbc = b->nblkcols;
  if (!group64_has_kernel(S, R) || nbr <= 0 || nbc <= 0) return 1;
  const int ng = (nbr + R - 1) / R, ngx = (ng + 7) / 8;
  const int64_t ngj = (int64_t)ng * nbc;
  if ((int64_t)ngx * nbc >= (1ll << 28)) return 1;
  if (!(reuse && E->group_built && E->group_R == R)) {
    if (E->group_flag.ensure(4) || E->groups.ensure((size_t)ngj * R + 1) || E->group_cnt.ensure((size_t)ngj + 1) || E->group_work.ensure((size_t)ngj + 1) ||
        E->group_start.ensure((size_t)ngj + 2))
      return -1;
    ACC_CHECK(hipMemsetAsync(E->group_flag.p, 0, sizeof(int), st));
    group_check_ascending(st, static_cast<const int64_t*>(b->blk_p), (int64_t)b->nblks, E->group_flag.p);
    int* hflag = reinterpret_cast<int*>(E->host_scalars + 12);
    ACC_CHECK(hipMemcpyAsync(hflag, E->group_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    ACC_CHECK(hipStreamSynchronize(st));
    E->b_monotone = *hflag == 0;
    E->group_R = R;
    E->group_built = true;
    if (E->b_monotone) {
      // the merged lists hold at most as many records as there are products
      if (E->group_entries.ensure((size_t)E->nproducts + 4)) return -1;
      group_build_table(st, c_out->row_p, c_out->col_i, E->descs.p, nbr, nbc, R, S, E->groups.p);
      group64_count(st, E->groups.p, E->descs.p, ngj, R, E->group_cnt.p);
      if (exclusive_scan<int64_t>(E, E->group_cnt.p, ngj, E->group_start.p, nullptr, true, st)) return -1;
      group64_merge(st, E->groups.p, E->descs.p, E->entries.p, ngj, R, S, E->group_start.p, E->group_work.p, E->group_entries.p);
    }
  }
  if (!E->b_monotone) return 1;
  GroupGeom G;
  G.nbc = nbc, G.ng = ng, G.ngx = ngx;
  const int64_t b_bytes = (int64_t)b->nblks * S * S * (int64_t)sizeof(double);
  const int64_t pb = E->group_panel_bytes > 0 ? E->group_panel_bytes : E->panel_bytes;
  int np = (int)std::min<int64_t>(std::max<int64_t>(1, (b_bytes + pb - 1) / pb), (int64_t)nbc);
  G.pw = (nbc + np - 1) / np;
  G.np = (nbc + G.pw - 1) / G.pw;
  const int has_tail = (E->min_k != E->max_k || E->max_k != S) ? 1 : 0;
  return group64_launch(S, R, st, E->descs.p, E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                        static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, skip_empty, has_tail, E->group_work.p,
                        E->group_entries.p, G);
}
#endif

static inline void plan_invalidate(Engine* E) { E->plan_saved = E->plan_hit = E->plan_numeric = false; }

// the twelve index arrays a plan depends on, as 32-bit words: patterns, block offsets and block sizes of A, B, C_in
static void plan_segments(const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in, const void* (&ptr)[12], long long (&n)[12]) {
  const void* p[12] = {a->row_p, a->col_i, a->blk_p, b->row_p, b->col_i, b->blk_p, c_in->row_p, c_in->col_i, c_in->blk_p,
                       a->row_blk_size, a->col_blk_size, b->col_blk_size};
  const long long w[12] = {a->nblkrows + 1ll, a->nblks, 2 * a->nblks, b->nblkrows + 1ll, b->nblks, 2 * b->nblks, c_in->nblkrows + 1ll, c_in->nblks,
                           2 * c_in->nblks, a->nblkrows, a->nblkcols, b->nblkcols};
  for (int i = 0; i < 12; ++i) ptr[i] = p[i], n[i] = w[i];
}

// 1 = the operands have exactly the index arrays of the saved plan (synchronises the stream once), 0 = not, < 0 error
static int plan_matches(Engine* E, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in, int retain, hipStream_t st) {
  if (!E->use_plan || !E->plan_saved) return 0;
  if (a->nblkrows != E->plan_dims[0] || a->nblkcols != E->plan_dims[1] || b->nblkcols != E->plan_dims[2] || a->nblks != E->plan_nblks[0] ||
      b->nblks != E->plan_nblks[1] || c_in->nblks != E->plan_nblks[2] || retain != E->plan_retain || E->canonical_c != E->plan_canonical)
    return 0;
  const void* ptr[12];
  long long n[12];
  plan_segments(a, b, c_in, ptr, n);
  if (E->plan_trusted && a->index_stamp && b->index_stamp && c_in->index_stamp && a->index_stamp == E->plan_stamps[0] &&
      b->index_stamp == E->plan_stamps[1] && c_in->index_stamp == E->plan_stamps[2]) {
    // same generation of the same arrays: the address test only guards against a caller that stamps carelessly
    bool same = true;
    for (int i = 0; i < 12; ++i) same = same && ptr[i] == E->plan_ptrs[i];
    if (same) return 1;
  }
  PlanSegs S;
  S.nseg = 12;
  long long off = 0, total = 0;
  for (int i = 0; i < 12; ++i) {
    S.a[i] = static_cast<const int32_t*>(ptr[i]);
    S.b[i] = E->plan_words.p + off;
    S.n[i] = n[i];
    off += n[i];
    total += n[i];
  }
  ACC_CHECK(hipMemsetAsync(E->plan_flag.p, 0, sizeof(int), st));
  const unsigned nb = (unsigned)std::min<long long>(2048, std::max<long long>(1, (total / 12 + 255) / 256));
  hipLaunchKernelGGL(plan_compare, dim3(nb), dim3(256), 0, st, S, E->plan_flag.p);
  ACC_CHECK(hipMemcpyAsync(E->plan_host_flag, E->plan_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  if (*E->plan_host_flag != 0) return 0;
  // equal arrays at (possibly) other addresses or of another generation: from now on THESE are the arrays the plan is known to fit, so a
  // caller that keeps them (a loop that passes its previous result back in) gets the cheap test next time
  for (int i = 0; i < 12; ++i) E->plan_ptrs[i] = ptr[i];
  E->plan_stamps[0] = a->index_stamp, E->plan_stamps[1] = b->index_stamp, E->plan_stamps[2] = c_in->index_stamp;
  return 1;
}

// keep device copies of the index arrays this symbolic phase saw, and of C's row pointer
static int plan_save(Engine* E, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in, int retain, const int32_t* c_row_p,
                     const dbcsr_amd_mm_counts& counts, hipStream_t st) {
  plan_invalidate(E);
  if (!E->use_plan) return 0;
  const void* ptr[12];
  long long n[12];
  plan_segments(a, b, c_in, ptr, n);
  long long total = 0;
  for (int i = 0; i < 12; ++i) total += n[i];
  const long long crow = a->nblkrows + 1ll;
  if (E->plan_words.ensure((size_t)(total + crow) + 1) || E->plan_flag.ensure(4)) return -1;
  long long off = 0;
  for (int i = 0; i < 12; ++i) {
    if (n[i] > 0) ACC_CHECK(hipMemcpyAsync(E->plan_words.p + off, ptr[i], sizeof(int32_t) * (size_t)n[i], hipMemcpyDeviceToDevice, st));
    off += n[i];
  }
  ACC_CHECK(hipMemcpyAsync(E->plan_words.p + off, c_row_p, sizeof(int32_t) * (size_t)crow, hipMemcpyDeviceToDevice, st));
  E->plan_dims[0] = a->nblkrows, E->plan_dims[1] = a->nblkcols, E->plan_dims[2] = b->nblkcols;
  E->plan_nblks[0] = a->nblks, E->plan_nblks[1] = b->nblks, E->plan_nblks[2] = c_in->nblks;
  E->plan_retain = retain;
  E->plan_canonical = E->canonical_c;
  E->plan_counts = counts;
  for (int i = 0; i < 12; ++i) E->plan_ptrs[i] = ptr[i];
  E->plan_stamps[0] = a->index_stamp, E->plan_stamps[1] = b->index_stamp, E->plan_stamps[2] = c_in->index_stamp;
  E->plan_saved = true;
  return 0;
}

#ifdef DBCSR_AMD_EXPERIMENTS
// The tile dataflow (mm_tile.h) for the C blocks of the dominant size: index work (bitmaps of A and of B transposed, sub-tile
// descriptors, k-sorted product lists), the persistent tile kernel, the products with inner blocks of another size.  The caller
// then runs the exact-size kernel over the C blocks of the other sizes.  descs[] and C_out's index are already filled.
template <int S_>
static int run_tile_f64(Engine* E, hipStream_t st, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
                        dbcsr_amd_bcsr* c_out, double alpha, double beta) {
  const int nbr = a->nblkrows, nbk = a->nblkcols, nbc = b->nblkcols, W = E->W, Wk = (nbk + 31) / 32;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    ACC_CHECK(hipGetDevice(&dev));
    ACC_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int cu_per_xcd = std::min(32, std::max(2, n_cu / 8));
  TileGeom G;
  G.nfr = E->hot_cnt_m;
  G.nfc = E->hot_cnt_n;
  if (!tile_shape(E->tile_shape, &G.tr, &G.tc, &G.wg_waves)) return 1;
  G.nTR = (G.nfr + G.tr - 1) / G.tr;
  G.nTC = (G.nfc + G.tc - 1) / G.tc;
  G.team_rows = std::max(1, cu_per_xcd * G.wg_waves / kTeamCols);
  G.nSR = (G.nTR + G.team_rows - 1) / G.team_rows;
  G.nSC = (G.nTC + kTeamCols - 1) / kTeamCols;
  G.nseq = (G.nSR * G.nSC + 7) / 8;
  G.kspan = nbk + 1;
  if ((int64_t)G.nseq * G.kspan >= 0x7ff00000ll) return 1;  // progress counter would overflow: not a tile case
  const int64_t nT = (int64_t)G.nTR * G.nTC;
  const bool reuse = E->plan_hit && E->plan_numeric && E->tile_built;
  if (E->tile_prog.ensure(8 * 256) || E->tile_flags.ensure(4)) return -1;
  ACC_CHECK(hipMemsetAsync(E->tile_prog.p, 0, sizeof(uint32_t) * 8 * 256, st));
  ACC_CHECK(hipMemsetAsync(E->tile_flags.p, 0, sizeof(int) * 4, st));
  if (!reuse) {
  if (E->a_bm.ensure((size_t)nbr * Wk + 1) || E->a_pre.ensure((size_t)nbr * Wk + 1) || E->bt_bm.ensure((size_t)nbc * Wk + 1) ||
      E->tile_rows.ensure((size_t)nbr + 1) || E->tile_cols.ensure((size_t)nbc + 1) || E->tdescs.ensure((size_t)nT + 1) ||
      E->tile_cnt.ensure((size_t)nT + 1) || E->tile_start.ensure((size_t)nT + 1) || E->tentries.ensure((size_t)E->nproducts + 1) ||
      false)
    return -1;
  ACC_CHECK(hipMemsetAsync(E->a_bm.p, 0, sizeof(uint32_t) * (size_t)nbr * Wk, st));
  ACC_CHECK(hipMemsetAsync(E->bt_bm.p, 0, sizeof(uint32_t) * (size_t)nbc * Wk, st));
  hipLaunchKernelGGL(bitmap_from_index, grid_for((int64_t)nbr * 64), dim3(256), 0, st, a->row_p, a->col_i, nbr, Wk, E->a_bm.p);
  hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->a_bm.p, nbr, Wk, E->a_pre.p, (int*)nullptr);
  hipLaunchKernelGGL(tile_bitmap_transposed, grid_for((int64_t)nbk * 64), dim3(256), 0, st, b->row_p, b->col_i, nbk, Wk, E->bt_bm.p);
  hipLaunchKernelGGL(tile_select, dim3(1), dim3(64), 0, st, a->row_blk_size, nbr, S_, E->tile_rows.p, nbr);
  hipLaunchKernelGGL(tile_select, dim3(1), dim3(64), 0, st, b->col_blk_size, nbc, S_, E->tile_cols.p, nbc);
  hipLaunchKernelGGL(tile_descs, grid_for(nT * 16), dim3(256), 0, st, G, E->tile_rows.p, E->tile_cols.p, E->c_bm.p, E->c_pre.p, c_out->row_p, W,
                     E->descs.p, E->tdescs.p, E->tile_cnt.p);
  if (exclusive_scan<int64_t>(E, E->tile_cnt.p, nT, E->tile_start.p, nullptr, false, st)) return -1;
  hipLaunchKernelGGL(tile_lists, grid_for(nT * 64), dim3(256), 0, st, G, E->tile_rows.p, E->tile_cols.p, nbk, Wk, E->a_bm.p, E->a_pre.p, a->row_p,
                     a->blk_p, E->bt_bm.p, W, E->b_bm.p, E->b_pre.p, b->row_p, b->blk_p, a->col_blk_size, S_, E->tile_start.p, E->tile_cnt.p,
                     E->tdescs.p, E->tentries.p, E->tile_flags.p + 1);
  E->tile_built = true;
  }
  TileArgs P;
  P.tdescs = E->tdescs.p;
  P.entries = E->tentries.p;
  P.a_data = static_cast<const double*>(a->data);
  P.b_data = static_cast<const double*>(b->data);
  P.c_out = static_cast<double*>(c_out->data);
  P.c_in = static_cast<const double*>(c_in->data);
  P.alpha = alpha;
  P.beta = beta;
  P.prog = E->tile_prog.p;
  P.flags = E->tile_flags.p;
  P.G = G;
  P.window = E->tile_window;
  P.pub_policy = E->tile_pub;
  P.prefetch = E->tile_prefetch;
  P.knobs = E->tile_knobs;
  P.times = nullptr;
  if (E->tile_knobs & 32) {
    if (E->tile_times.ensure(8)) return -1;
    ACC_CHECK(hipMemsetAsync(E->tile_times.p, 0, 8 * sizeof(unsigned long long), st));
    P.times = E->tile_times.p;
  }
  ACC_CHECK(hipEventRecord(E->ev[1], st));  // the timed numeric launch starts here (the index work above counts as fill time)
  if (tile_launch(S_, S_, S_, E->tile_rdv, E->tile_shape, (unsigned)(8 * cu_per_xcd), st, P)) return -1;
  if (tile_launch_remainder(S_, S_, st, G, E->tdescs.p, E->tentries.p, P.a_data, P.b_data, P.c_out, alpha)) return -1;
  return check(hipGetLastError(), "run_tile_f64", __FILE__, __LINE__);
}

// The band dataflow (mm_band.h) for the C blocks of the dominant size: bitmaps of A and of B transposed, sub-tile descriptors, the
// product lists in sweep order (count, scan, fill), the persistent kernel, the products with inner blocks of another size.  The
// caller then runs the exact-size kernel over the C blocks of the other sizes.  descs[] and C_out's index are already filled.
template <int S_>
static int run_band_f64(Engine* E, hipStream_t st, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
                        dbcsr_amd_bcsr* c_out, double alpha, double beta) {
  const int nbr = a->nblkrows, nbk = a->nblkcols, nbc = b->nblkcols, W = E->W, Wk = (nbk + 31) / 32;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    ACC_CHECK(hipGetDevice(&dev));
    ACC_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  if (band_lds_bytes(S_, S_, S_, E->band_shape, E->band_depth) == 0) return 1;
  BandGeom G;
  G.nfr = E->hot_cnt_m;
  G.nfc = E->hot_cnt_n;
  if (G.nfr <= 0 || G.nfc <= 0 || !band_shape(E->band_shape, &G.waves, &G.tr, &G.tc)) return 1;
  G.nBR = (G.nfr + G.waves * G.tr - 1) / (G.waves * G.tr);
  G.nBC = (G.nfc + G.tc - 1) / G.tc;
  if ((int64_t)G.nBR * G.nBC > 0x3fffffff) return 1;
  G.ntiles = G.nBR * G.nBC;
  G.cu_per_xcd = std::min(32, std::max(1, n_cu / 8));
  G.max_i = 1;
  for (int x = 0; x < 8; ++x) G.max_i = std::max(G.max_i, (int)((G.lo(x + 1) - G.lo(x) + G.cu_per_xcd - 1) / G.cu_per_xcd));
  G.kshift = 0;
  while ((nbk >> G.kshift) >= 4096) ++G.kshift;
  G.kspan = (nbk >> G.kshift) + 1;
  if ((int64_t)(G.max_i + 1) * G.kspan >= 0x7ff00000ll) return 1;  // sweep positions would overflow: not a band case
  const int nwg = 8 * G.cu_per_xcd;
  const int64_t nsub = (int64_t)G.waves * G.ntiles, npl = (int64_t)nwg * G.waves * G.max_i, nps = (int64_t)nwg * G.max_i;
  const bool reuse = E->plan_hit && E->plan_numeric && E->band_built && E->band_geom.waves == G.waves && E->band_geom.ntiles == G.ntiles;
  if (E->band_flags.ensure(4) || E->band_prog.ensure(8 * 512)) return -1;
  ACC_CHECK(hipMemsetAsync(E->band_flags.p, 0, sizeof(int) * 4, st));
  ACC_CHECK(hipMemsetAsync(E->band_prog.p, 0, sizeof(unsigned) * 8 * 512, st));
  if (!reuse) {
    E->band_built = false;
    if (E->a_bm.ensure((size_t)nbr * Wk + 1) || E->a_pre.ensure((size_t)nbr * Wk + 1) || E->bt_bm.ensure((size_t)nbc * Wk + 1) ||
        E->tile_rows.ensure((size_t)nbr + 1) || E->tile_cols.ensure((size_t)nbc + 1) || E->band_descs_buf.ensure((size_t)nsub + 1) ||
        E->band_sub_cnt.ensure((size_t)nsub + 1) || E->band_cnt_rem.ensure((size_t)nsub + 1) || E->band_rem_start.ensure((size_t)nsub + 2) ||
        E->band_cnt_list.ensure((size_t)npl + 1) || E->band_list_off.ensure((size_t)npl + 2) || E->band_cnt_b.ensure((size_t)nps + 1) ||
        E->band_seq_off.ensure((size_t)nps + 2))
      return -1;
    ACC_CHECK(hipMemsetAsync(E->a_bm.p, 0, sizeof(uint32_t) * (size_t)nbr * Wk, st));
    ACC_CHECK(hipMemsetAsync(E->bt_bm.p, 0, sizeof(uint32_t) * (size_t)nbc * Wk, st));
    ACC_CHECK(hipMemsetAsync(E->band_cnt_list.p, 0, sizeof(int) * (size_t)npl, st));
    ACC_CHECK(hipMemsetAsync(E->band_cnt_b.p, 0, sizeof(int) * (size_t)nps, st));
    hipLaunchKernelGGL(bitmap_from_index, grid_for((int64_t)nbr * 64), dim3(256), 0, st, a->row_p, a->col_i, nbr, Wk, E->a_bm.p);
    hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->a_bm.p, nbr, Wk, E->a_pre.p, (int*)nullptr);
    hipLaunchKernelGGL(tile_bitmap_transposed, grid_for((int64_t)nbk * 64), dim3(256), 0, st, b->row_p, b->col_i, nbk, Wk, E->bt_bm.p);
    hipLaunchKernelGGL(tile_select, dim3(1), dim3(64), 0, st, a->row_blk_size, nbr, S_, E->tile_rows.p, nbr);
    hipLaunchKernelGGL(tile_select, dim3(1), dim3(64), 0, st, b->col_blk_size, nbc, S_, E->tile_cols.p, nbc);
    hipLaunchKernelGGL(band_descs, grid_for(nsub * 16), dim3(256), 0, st, G, E->tile_rows.p, E->tile_cols.p, E->c_bm.p, E->c_pre.p, c_out->row_p, W,
                       E->descs.p, E->band_descs_buf.p, E->band_sub_cnt.p);
    hipLaunchKernelGGL((band_lists<false>), grid_for(nsub * 64), dim3(256), 0, st, G, E->tile_rows.p, E->tile_cols.p, nbk, Wk, E->a_bm.p, E->a_pre.p,
                       a->row_p, a->blk_p, E->bt_bm.p, W, E->b_bm.p, E->b_pre.p, b->row_p, b->blk_p, a->col_blk_size, S_, E->band_cnt_list.p,
                       E->band_cnt_b.p, E->band_cnt_rem.p, (const int64_t*)nullptr, (const int64_t*)nullptr, (const int64_t*)nullptr,
                       E->band_sub_cnt.p, (BandEntry*)nullptr, (BandRem*)nullptr, E->band_flags.p + 1);
    if (exclusive_scan<int64_t>(E, E->band_cnt_list.p, npl, E->band_list_off.p, nullptr, true, st)) return -1;
    if (exclusive_scan<int64_t>(E, E->band_cnt_b.p, nps, E->band_seq_off.p, nullptr, true, st)) return -1;
    if (exclusive_scan<int64_t>(E, E->band_cnt_rem.p, nsub, E->band_rem_start.p, nullptr, true, st)) return -1;
    hipLaunchKernelGGL(band_max_seq, grid_for(nwg), dim3(256), 0, st, G, nwg, E->band_seq_off.p, E->band_flags.p + 2);
    // list sizes to the host (once per plan: a multiply that reuses the plan comes nowhere near this)
    ACC_CHECK(hipMemcpyAsync(E->host_scalars + 8, E->band_list_off.p + npl, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    ACC_CHECK(hipMemcpyAsync(E->host_scalars + 9, E->band_rem_start.p + nsub, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    ACC_CHECK(hipMemcpyAsync(E->host_scalars + 10, E->band_flags.p + 2, sizeof(int), hipMemcpyDeviceToHost, st));
    ACC_CHECK(hipStreamSynchronize(st));
    E->band_nlist = E->host_scalars[8];
    E->band_nrem = E->host_scalars[9];
    const int max_seq = *reinterpret_cast<const int*>(E->host_scalars + 10);
    if (max_seq >= (1 << 23) - 64) return 1;  // the entries carry 23 bits of the sequence number: not a band case
    if (E->band_entries.ensure((size_t)E->band_nlist + 1) || E->band_rem.ensure((size_t)E->band_nrem + 1)) return -1;
    hipLaunchKernelGGL((band_lists<true>), grid_for(nsub * 64), dim3(256), 0, st, G, E->tile_rows.p, E->tile_cols.p, nbk, Wk, E->a_bm.p, E->a_pre.p,
                       a->row_p, a->blk_p, E->bt_bm.p, W, E->b_bm.p, E->b_pre.p, b->row_p, b->blk_p, a->col_blk_size, S_, (int*)nullptr,
                       (int*)nullptr, (int*)nullptr, E->band_list_off.p, E->band_seq_off.p, E->band_rem_start.p, E->band_sub_cnt.p,
                       E->band_entries.p, E->band_rem.p, E->band_flags.p + 1);
    E->band_geom = G;
    E->band_built = true;
  }
  BandArgs P;
  P.descs = E->band_descs_buf.p;
  P.entries = E->band_entries.p;
  P.list_off = E->band_list_off.p;
  P.a_data = static_cast<const double*>(a->data);
  P.b_data = static_cast<const double*>(b->data);
  P.c_out = static_cast<double*>(c_out->data);
  P.c_in = static_cast<const double*>(c_in->data);
  P.alpha = alpha;
  P.beta = beta;
  P.G = G;
  P.flags = E->band_flags.p;
  P.prog = E->band_prog.p;
  P.window = E->band_window > 0 ? std::max(1, E->band_window >> G.kshift) : 0;
  P.knobs = E->band_knobs;
  P.times = nullptr;
  if (E->band_knobs & 1) {
    if (E->band_times.ensure(16)) return -1;
    ACC_CHECK(hipMemsetAsync(E->band_times.p, 0, 16 * sizeof(unsigned long long), st));
    P.times = E->band_times.p;
  }
  ACC_CHECK(hipEventRecord(E->ev[1], st));  // the timed numeric launch starts here (the index work above counts as fill time)
  if (band_launch(S_, S_, S_, E->band_shape, E->band_depth, E->band_bpol, (unsigned)nwg, st, P)) return -1;
  if (E->band_nrem > 0 &&
      band_launch_remainder(S_, S_, st, nsub, E->band_descs_buf.p, E->band_rem_start.p, E->band_rem.p, P.a_data, P.b_data, P.c_out, alpha))
    return -1;
  return check(hipGetLastError(), "run_band_f64", __FILE__, __LINE__);
}
#endif  // DBCSR_AMD_EXPERIMENTS

}  // namespace dbcsr_amd

using namespace dbcsr_amd;

extern "C" {

int dbcsr_amd_mm_create(void** handle) {
  if (!handle) return -1;
  Engine* E = new (std::nothrow) Engine();
  if (!E) return -1;
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&E->host_scalars), 16 * sizeof(int64_t), hipHostMallocDefault);
  if (e != hipSuccess) {
    delete E;
    return check(e, "hipHostMalloc", __FILE__, __LINE__);
  }
  if (const char* k = getenv("DBCSR_AMD_MM_KERNEL")) {
    E->use_lds = strcmp(k, "direct") != 0;
    E->use_pipe = strcmp(k, "pipe") == 0 ? 1 : (strcmp(k, "lds1") == 0 ? 0 : -1);
#ifdef DBCSR_AMD_EXPERIMENTS
    if (strncmp(k, "dma", 3) == 0 && k[3] >= '2' && k[3] <= '4') E->dma_stages = k[3] - '0';
#endif
  }
  if (const char* k = getenv("DBCSR_AMD_MM_PIPE_G")) E->pipe_g = std::max(1, atoi(k));
  if (const char* k = getenv("DBCSR_AMD_MM_CLASSES")) E->use_classes = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_WORK")) E->use_work = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_WG_WAVES")) {
    const int w = atoi(k);
    if (w == 1 || w == 2 || w == 4) E->wg_waves = w;
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&E->cls_host_hist), 3 * 33 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&E->cls_host_lens), (2 * kNumClasses + 1) * sizeof(int64_t), hipHostMallocDefault) != hipSuccess)
    return -1;
  if (const char* k = getenv("DBCSR_AMD_MM_PLAN")) E->use_plan = atoi(k);
  if (hipHostMalloc(reinterpret_cast<void**>(&E->plan_host_flag), sizeof(int), hipHostMallocDefault) != hipSuccess) return -1;
  if (const char* k = getenv("DBCSR_AMD_MM_HOT")) E->use_hot = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TINY")) E->use_tiny = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_F32_DIRECT")) E->f32_direct = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_BIG")) E->use_big = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_MID")) E->use_mid = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_F32_GROUP")) {
    const int r = atoi(k);
    E->f32_group = (r >= 2 && r <= 4) ? r : (r < 0 ? -1 : 0);
  }
  if (const char* k = getenv("DBCSR_AMD_MM_F64_GROUP")) E->f64_group = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_GROUP_PANEL_MB")) E->group_panel_bytes = (int64_t)atoll(k) << 20;
  if (const char* k = getenv("DBCSR_AMD_MM_SYMBOLIC")) {
    E->force_word_kernels = strcmp(k, "word") == 0;
    E->force_symbolic = strcmp(k, "word") == 0 ? 1 : (strcmp(k, "grid") == 0 ? 2 : (strcmp(k, "rows") == 0 ? 3 : 0));
  }
  if (const char* k = getenv("DBCSR_AMD_MM_PANEL_MB")) E->panel_bytes = (int64_t)atoll(k) << 20;
#ifdef DBCSR_AMD_EXPERIMENTS
  // ---- the lab build's switches (every one selects something that was measured and does not win; see the top of this file) ----
  if (const char* k = getenv("DBCSR_AMD_MM_CLASS_G")) {
    const int g = atoi(k);
    E->class_g = (g == 2 || g == 4 || g == 8) ? g : 1;
  }
  if (const char* k = getenv("DBCSR_AMD_MM_DBG")) E->dbg = atoi(k);
  {
    const char* k = getenv("DBCSR_AMD_MM_POISON");  // (process-wide: the engines created from now on)
    g_devbuf_poison = k ? (atoi(k) & 255) : -1;
  }
  if (const char* k = getenv("DBCSR_AMD_MM_HOT_VARIANT")) E->hot_variant = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_HOT_PERSISTENT")) E->hot_persistent = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_HOT_XCDS")) E->hot_xcd_mask = (unsigned)strtoul(k, nullptr, 0) & 0xffu;
  if (const char* k = getenv("DBCSR_AMD_MM_TILE")) E->use_tile = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_WINDOW")) E->tile_window = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_RDV")) E->tile_rdv = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_PUB")) E->tile_pub = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_PREFETCH")) E->tile_prefetch = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_KNOBS")) E->tile_knobs = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_SHAPE")) E->tile_shape = atoi(k) == 1 ? 1 : 0;
  if (const char* k = getenv("DBCSR_AMD_MM_BAND")) E->use_band = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_BAND_DEPTH")) E->band_depth = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_BAND_BPOL")) E->band_bpol = atoi(k) == 1 ? 1 : 0;
  if (const char* k = getenv("DBCSR_AMD_MM_BAND_KNOBS")) E->band_knobs = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_BAND_WINDOW")) E->band_window = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_BAND_SHAPE")) E->band_shape = atoi(k) == 0 ? 0 : 1;
  if (const char* k = getenv("DBCSR_AMD_MM_LDS_PAD")) E->lds_pad = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_CLASS_STREAMS")) E->class_streams = std::min(4, std::max(1, atoi(k)));
  if (const char* k = getenv("DBCSR_AMD_MM_ROW_GROUP")) E->row_group = atoi(k);
#endif
  for (int i = 0; i < 3; ++i) {
    e = hipEventCreate(&E->ev[i]);
    if (e != hipSuccess) return check(e, "hipEventCreate", __FILE__, __LINE__);
  }
  *handle = E;
  return 0;
}

int dbcsr_amd_mm_timing(void* handle, float* ms_fill, float* ms_numeric) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !E->timed) return -1;
  ACC_CHECK(hipEventSynchronize(E->ev[2]));
  float f = 0.f, n = 0.f;
  ACC_CHECK(hipEventElapsedTime(&f, E->ev[0], E->ev[1]));
  ACC_CHECK(hipEventElapsedTime(&n, E->ev[1], E->ev[2]));
  if (ms_fill) *ms_fill = f;
  if (ms_numeric) *ms_numeric = n;
  return 0;
}

int dbcsr_amd_mm_destroy(void* handle) {
  if (!handle) return 0;
  Engine* E = static_cast<Engine*>(handle);
  E->b_bm.release(); E->c_bm.release(); E->cin_bm.release();
  E->b_pre.release(); E->c_pre.release(); E->cin_pre.release(); E->row_nnz.release(); E->prod_cnt.release();
  E->blk_nze.release(); E->tmp_i32.release();
  E->prod_start.release(); E->c_blk_p_ws.release(); E->partial.release(); E->off_a.release(); E->off_b.release();
  E->entries.release(); E->descs.release(); E->row_sums.release(); E->dev_scalars.release();
  E->order.release(); E->order_cnt.release(); E->order_base.release();
  E->stat_table.release();
  E->norms64.release(); E->a_norms.release(); E->b_norms.release(); E->keep.release();
  E->hot_counters.release();
  E->a_bm.release(); E->bt_bm.release(); E->tile_prog.release(); E->a_pre.release(); E->tile_rows.release(); E->tile_cols.release();
  E->tile_cnt.release(); E->tile_flags.release(); E->tile_start.release();
#ifdef DBCSR_AMD_EXPERIMENTS
  E->tdescs.release(); E->tentries.release(); E->band_descs_buf.release(); E->band_entries.release(); E->band_rem.release();
#endif
  E->band_cnt_list.release(); E->band_cnt_b.release();
  E->band_cnt_rem.release(); E->band_sub_cnt.release(); E->band_flags.release(); E->band_list_off.release(); E->band_seq_off.release();
  E->band_rem_start.release(); E->band_times.release(); E->band_prog.release();
  if (E->host_scalars) (void)hipHostFree(E->host_scalars);
  if (E->plan_host_flag) (void)hipHostFree(E->plan_host_flag);
  E->plan_words.release(); E->plan_c_col_i.release(); E->plan_c_blk_p.release(); E->plan_flag.release(); E->work.release();
  if (E->cls_host_hist) (void)hipHostFree(E->cls_host_hist);
  if (E->cls_host_lens) (void)hipHostFree(E->cls_host_lens);
  E->cls_hist.release(); E->cls_row.release(); E->cls_col.release(); E->cls_col_bm.release(); E->cls_lens.release();
  for (int i = 0; i < 3; ++i)
    if (E->ev[i]) (void)hipEventDestroy(E->ev[i]);
  for (int i = 0; i < 3; ++i) {
    if (E->side_stream[i]) (void)hipStreamDestroy(E->side_stream[i]);
    if (E->join_ev[i]) (void)hipEventDestroy(E->join_ev[i]);
  }
  if (E->fork_ev) (void)hipEventDestroy(E->fork_ev);
  delete E;
  return 0;
}

int dbcsr_amd_mm_symbolic(void* handle, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
                          int retain_sparsity, int32_t* c_out_row_p, dbcsr_amd_mm_counts* counts, void* stream) {
  return dbcsr_amd_mm_symbolic_filtered(handle, dbcsr_type_real_8, 1.0, 0.0, a, b, c_in, retain_sparsity, c_out_row_p, counts, stream);
}

int dbcsr_amd_mm_symbolic_filtered(void* handle, libsmm_acc_data_t datatype, double alpha, double filter_eps, const dbcsr_amd_bcsr* a,
                                   const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in, int retain_sparsity, int32_t* c_out_row_p,
                                   dbcsr_amd_mm_counts* counts, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !a || !b || !c_in || !c_out_row_p || !counts) return -1;
  const bool filtering = filter_eps > 0.0;
  if (filtering && datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  if (a->nblkcols != b->nblkrows || a->nblkrows != c_in->nblkrows || b->nblkcols != c_in->nblkcols) {
    fprintf(stderr, "dbcsr_amd_mm_symbolic: incompatible block dimensions\n");
    return -2;
  }
  hipStream_t st = stream_of(stream);
  const int nbr = a->nblkrows, nbk = a->nblkcols, nbc = b->nblkcols;
  const int W = (nbc + 31) / 32;
  // same index arrays as the previous multiply of this engine: its plan stands (no on-the-fly filter: that one depends on the values)
  if (!filtering) {
    const int hit = plan_matches(E, a, b, c_in, retain_sparsity ? 1 : 0, st);
    if (hit < 0) return -1;
    if (hit) {
      long long off = 0;
      const void* ptr[12];
      long long n[12];
      plan_segments(a, b, c_in, ptr, n);
      for (int i = 0; i < 12; ++i) off += n[i];
      ACC_CHECK(hipMemcpyAsync(c_out_row_p, E->plan_words.p + off, sizeof(int32_t) * ((size_t)nbr + 1), hipMemcpyDeviceToDevice, st));
      *counts = E->plan_counts;
      E->norms_data = nullptr;
      E->filter = FilterArgs{nullptr, nullptr, 0.0f};
      E->valid = true;
      E->plan_hit = true;
      ++E->plan_hits;
      return 0;
    }
  }
  plan_invalidate(E);
  E->valid = false;
  E->nbr = nbr;
  E->W = W;
  E->retain = retain_sparsity != 0;
  E->norms_data = nullptr;  // block norms left by an earlier numeric phase belong to that product only
  E->have_cin = c_in->nblks > 0;
  if (E->b_bm.ensure((size_t)nbk * W + 1) || E->b_pre.ensure((size_t)nbk * W + 1) || E->c_bm.ensure((size_t)nbr * W + 1) ||
      E->c_pre.ensure((size_t)nbr * W + 1) || E->row_nnz.ensure((size_t)nbr + 1) || E->dev_scalars.ensure(16))
    return -1;
  if (E->have_cin && (E->cin_bm.ensure((size_t)nbr * W + 1) || E->cin_pre.ensure((size_t)nbr * W + 1))) return -1;
  ACC_CHECK(hipMemsetAsync(E->dev_scalars.p, 0, 16 * sizeof(unsigned long long), st));
  {  // the "negated min" slots start at the most negative value
    static const int init[6] = {0, -0x7fffffff, 0, -0x7fffffff, 0, -0x7fffffff};
    ACC_CHECK(hipMemcpyAsync(E->dev_scalars.p + 4, init, sizeof(init), hipMemcpyHostToDevice, st));
  }
  if (nbr == 0 || nbc == 0) {
    ACC_CHECK(hipMemsetAsync(c_out_row_p, 0, sizeof(int32_t) * ((size_t)nbr + 1), st));
    ACC_CHECK(hipStreamSynchronize(st));
    counts->c_nblks = counts->c_nze = counts->nproducts = counts->flop = 0;
    E->c_nblks = 0;
    E->nproducts = 0;
    E->valid = true;
    return 0;
  }
  // 1. bitmaps of B (and C_in)
  ACC_CHECK(hipMemsetAsync(E->b_bm.p, 0, sizeof(uint32_t) * (size_t)nbk * W, st));
  if (nbk > 0) {
    hipLaunchKernelGGL(bitmap_from_index, grid_for((int64_t)nbk * 64), dim3(256), 0, st, b->row_p, b->col_i, nbk, W, E->b_bm.p);
    hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbk * 64), dim3(256), 0, st, E->b_bm.p, nbk, W, E->b_pre.p, (int*)nullptr);
  }
  if (E->have_cin) {
    ACC_CHECK(hipMemsetAsync(E->cin_bm.p, 0, sizeof(uint32_t) * (size_t)nbr * W, st));
    hipLaunchKernelGGL(bitmap_from_index, grid_for((int64_t)nbr * 64), dim3(256), 0, st, c_in->row_p, c_in->col_i, nbr, W,
                       E->cin_bm.p);
    hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->cin_bm.p, nbr, W, E->cin_pre.p, (int*)nullptr);
  }
  // on-the-fly filter: block norms of A and alpha*B (fp32 values of fp64 sums)
  E->filter = FilterArgs{nullptr, nullptr, 0.0f};
  if (filtering) {
    if (E->a_norms.ensure((size_t)a->nblks + 1) || E->b_norms.ensure((size_t)b->nblks + 1)) return -1;
    const int sa = row_split(nbr, a->nblks), sb = row_split(nbk, b->nblks);
    if (datatype == dbcsr_type_real_8) {
      hipLaunchKernelGGL((bcsr_block_norms<double>), grid_for((int64_t)nbr * sa * 64), dim3(256), 0, st, a->row_p, a->col_i, a->blk_p,
                         static_cast<const double*>(a->data), a->row_blk_size, a->col_blk_size, nbr, sa, 1.0, E->a_norms.p, (double*)nullptr);
      hipLaunchKernelGGL((bcsr_block_norms<double>), grid_for((int64_t)nbk * sb * 64), dim3(256), 0, st, b->row_p, b->col_i, b->blk_p,
                         static_cast<const double*>(b->data), b->row_blk_size, b->col_blk_size, nbk, sb, alpha, E->b_norms.p, (double*)nullptr);
    } else {
      hipLaunchKernelGGL((bcsr_block_norms<float>), grid_for((int64_t)nbr * sa * 64), dim3(256), 0, st, a->row_p, a->col_i, a->blk_p,
                         static_cast<const float*>(a->data), a->row_blk_size, a->col_blk_size, nbr, sa, 1.0, E->a_norms.p, (double*)nullptr);
      hipLaunchKernelGGL((bcsr_block_norms<float>), grid_for((int64_t)nbk * sb * 64), dim3(256), 0, st, b->row_p, b->col_i, b->blk_p,
                         static_cast<const float*>(b->data), b->row_blk_size, b->col_blk_size, nbk, sb, alpha, E->b_norms.p, (double*)nullptr);
    }
    E->filter = FilterArgs{E->a_norms.p, E->b_norms.p, (float)filter_eps};
  }
  // 2. pattern of C_out, its row prefix and row pointer
  // expected number of products against the number of (row, column) candidates: product-driven kernels for a sparse product
  const double prod_est = (double)a->nblks * ((double)b->nblks / (double)std::max(nbk, 1));
  const bool sparse_guess = E->force_symbolic == 3 || (E->force_symbolic == 0 && nbr >= 2048 && prod_est < 0.6 * (double)nbr * (double)nbc);
  if (filtering && sparse_guess && !retain_sparsity) {
    if (E->have_cin)
      ACC_CHECK(hipMemcpyAsync(E->c_bm.p, E->cin_bm.p, sizeof(uint32_t) * (size_t)nbr * W, hipMemcpyDeviceToDevice, st));
    else
      ACC_CHECK(hipMemsetAsync(E->c_bm.p, 0, sizeof(uint32_t) * (size_t)nbr * W, st));
    hipLaunchKernelGGL(c_bitmap_rows_filtered, grid_for((int64_t)nbr * 64), dim3(256), 0, st, a->row_p, a->col_i, b->row_p, b->col_i, nbr, W,
                       E->canonical_c, E->filter, E->c_bm.p);
  } else if (filtering)
    hipLaunchKernelGGL(c_bitmap_filtered, grid_for((int64_t)nbr * ((nbc + 63) / 64) * 64), dim3(256), 0, st, a->row_p, a->col_i, b->row_p,
                       E->b_bm.p, E->b_pre.p, E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr, nbr, nbc, W, (nbc + 63) / 64,
                       retain_sparsity ? 1 : 0, E->canonical_c, E->filter, E->c_bm.p);
  else
    hipLaunchKernelGGL(c_bitmap, grid_for((int64_t)nbr * W), dim3(256), 0, st, a->row_p, a->col_i, E->b_bm.p,
                       E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr, nbr, W, retain_sparsity ? 1 : 0, E->canonical_c, E->c_bm.p);
  hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->c_bm.p, nbr, W, E->c_pre.p, E->row_nnz.p);
  int64_t* dsc = reinterpret_cast<int64_t*>(E->dev_scalars.p);
  if (exclusive_scan<int32_t>(E, E->row_nnz.p, nbr, c_out_row_p, dsc + 0, true, st)) return -1;
  // block-size maxima (LDS slice size / kernel choice of the numeric phase)
  {
    int* mx = reinterpret_cast<int*>(E->dev_scalars.p + 4);
    hipLaunchKernelGGL(max_of, dim3(1), dim3(256), 0, st, a->row_blk_size, nbr, mx + 0);
    hipLaunchKernelGGL(max_of, dim3(1), dim3(256), 0, st, a->col_blk_size, nbk, mx + 2);
    hipLaunchKernelGGL(max_of, dim3(1), dim3(256), 0, st, b->col_blk_size, nbc, mx + 4);
    // ... and the most frequent block size per dimension (choice of an exact-size kernel)
    int* md = reinterpret_cast<int*>(E->dev_scalars.p + 8);
    hipLaunchKernelGGL(mode_of, dim3(1), dim3(256), 0, st, a->row_blk_size, nbr, md + 0);
    hipLaunchKernelGGL(mode_of, dim3(1), dim3(256), 0, st, a->col_blk_size, nbk, md + 2);
    hipLaunchKernelGGL(mode_of, dim3(1), dim3(256), 0, st, b->col_blk_size, nbc, md + 4);
  }
  // block-size histograms (sizes 1..32) of the three dimensions: the (m, n) classes of a mixed-size multiply
  E->cls_mode = false;
  if (E->use_classes > 0) {
    if (E->cls_hist.ensure(3 * 33)) return -1;
    ACC_CHECK(hipMemsetAsync(E->cls_hist.p, 0, 3 * 33 * sizeof(int), st));
    hipLaunchKernelGGL(size_hist, grid_for(nbr), dim3(256), 0, st, a->row_blk_size, nbr, E->cls_hist.p);
    hipLaunchKernelGGL(size_hist, grid_for(nbc), dim3(256), 0, st, b->col_blk_size, nbc, E->cls_hist.p + 33);
    hipLaunchKernelGGL(size_hist, grid_for(nbk), dim3(256), 0, st, a->col_blk_size, nbk, E->cls_hist.p + 66);
    ACC_CHECK(hipMemcpyAsync(E->cls_host_hist, E->cls_hist.p, 3 * 33 * sizeof(int), hipMemcpyDeviceToHost, st));
  }
  // need c_nblks (and the block-size extrema) on the host to size per-block work arrays
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 11 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  const int64_t c_nblks = E->host_scalars[0];
  {
    const int* mx = reinterpret_cast<const int*>(E->host_scalars + 4);
    E->max_m = mx[0]; E->min_m = -mx[1];
    E->max_k = mx[2]; E->min_k = -mx[3];
    E->max_n = mx[4]; E->min_n = -mx[5];
    if (E->max_k > 0xffff || E->max_m > 0x7fff || E->max_n > 0x7fff) {
      fprintf(stderr, "dbcsr_amd_mm_symbolic: block sizes above 32767 (m, n) / 65535 (k) are not supported (packed 16-bit extents)\n");
      return -1;
    }
    // exact-size kernel: only when one (m, n, k) covers at least 90 % of the block rows / columns of each dimension
    const int* md = reinterpret_cast<const int*>(E->host_scalars + 8);
    const bool dominant = 10ll * md[1] >= 9ll * nbr && 10ll * md[3] >= 9ll * nbk && 10ll * md[5] >= 9ll * nbc;
    E->hot_m = dominant ? md[0] : 0;
    E->hot_k = dominant ? md[2] : 0;
    E->hot_n = dominant ? md[4] : 0;
    E->hot_cnt_m = md[1], E->hot_cnt_k = md[3], E->hot_cnt_n = md[5];
  }
  // (m, n) classes: blocks of at most 32 in every dimension, no single dominant size (that case has its ahead-of-time
  // kernel), not the packed 4 x 4 case, and enough C blocks to pay for compiling the class kernels (forced with
  // DBCSR_AMD_MM_CLASSES=2)
  if (E->use_classes > 0 && E->max_m <= 32 && E->max_k <= 32 && E->max_n <= 32 && E->min_m >= 1 && E->min_k >= 1 && E->min_n >= 1 &&
      !(E->max_m <= 4 && E->max_n <= 4) && (E->use_classes > 1 || (E->hot_m == 0 && c_nblks >= 200000))) {
    auto top3 = [](const int* hist, int* out) {
      int used[3] = {-1, -1, -1};
      for (int r = 0; r < 3; ++r) {
        int best = 0, bc = 0;
        for (int sz = 1; sz <= 32; ++sz)
          if (hist[sz] > bc && sz != used[0] && sz != used[1]) best = sz, bc = hist[sz];
        out[r] = best;
        used[r] = best ? best : -1;
      }
    };
    top3(E->cls_host_hist, E->cls_m);
    top3(E->cls_host_hist + 33, E->cls_n);
    top3(E->cls_host_hist + 66, E->cls_k);
    E->cls_mode = E->cls_m[0] > 0 && E->cls_n[0] > 0 && E->cls_k[0] > 0;
  }
  // processing order of the numeric phase: column panels sized for the Infinity Cache, rows dealt to XCDs
  // size of B from the mean block sizes when the histograms are at hand (mixed sizes: the maxima overestimate it 2x on
  // BASELINE config 3, which doubled the number of panels and with it the compulsory re-reads of the A block-rows)
  double mean_k = E->max_k, mean_n = E->max_n;
  if (E->use_classes > 0 && E->max_k <= 32 && E->max_n <= 32 && E->min_k >= 1 && E->min_n >= 1) {
    double sk = 0, ck = 0, sn = 0, cn = 0;
    for (int sz = 1; sz <= 32; ++sz) {
      sn += (double)sz * E->cls_host_hist[33 + sz];
      cn += E->cls_host_hist[33 + sz];
      sk += (double)sz * E->cls_host_hist[66 + sz];
      ck += E->cls_host_hist[66 + sz];
    }
    if (ck > 0 && cn > 0) mean_k = sk / ck, mean_n = sn / cn;
  }
  const int64_t b_bytes_est = (int64_t)((double)b->nblks * mean_k * mean_n * (double)sizeof(double));
  int NP = (int)std::min<int64_t>((b_bytes_est + E->panel_bytes - 1) / E->panel_bytes, (int64_t)W);
  if (NP < 1) NP = 1;
  const int PW = (W + NP - 1) / NP;
  NP = (W + PW - 1) / PW;
  const int R = (nbr + 7) / 8;
  // rows walked together per XCD.  Measured on config 2 (DBCSR_AMD_MM_ROW_GROUP = 1/2/4/6/8: 22.7/22.8/23.5/25.2/26.3 ms):
  // the B reuse it buys (10 % fill: 14 % fewer B fetches at 4 rows) does not pay for the extra A rows in L2 -> default 1.
  int RG = E->row_group > 0 ? E->row_group : 1;
  RG = std::max(1, std::min(RG, R));
  const int NG = (R + RG - 1) / RG;
  const int nkeys = (E->cls_mode ? kNumClasses : 1) * 8 * NP * (E->cls_mode ? R : NG);
  if (E->order_cnt.ensure((size_t)nkeys + 1) || E->order_base.ensure((size_t)nkeys + 1)) return -1;
  if (E->cls_mode) {
    if (E->cls_row.ensure((size_t)nbr + 1) || E->cls_col.ensure((size_t)nbc + 1) || E->cls_col_bm.ensure((size_t)4 * W + 1) ||
        E->cls_lens.ensure(2 * kNumClasses + 1))
      return -1;
    hipLaunchKernelGGL(class_ids, grid_for(nbr), dim3(256), 0, st, a->row_blk_size, nbr, E->cls_m[0], E->cls_m[1], E->cls_m[2], E->cls_row.p);
    hipLaunchKernelGGL(class_ids, grid_for(nbc), dim3(256), 0, st, b->col_blk_size, nbc, E->cls_n[0], E->cls_n[1], E->cls_n[2], E->cls_col.p);
    hipLaunchKernelGGL(class_col_bitmaps, grid_for(W), dim3(256), 0, st, E->cls_col.p, nbc, W, E->cls_col_bm.p);
    hipLaunchKernelGGL(order_count_cls, grid_for(nkeys), dim3(256), 0, st, E->c_bm.p, E->cls_row.p, E->cls_col_bm.p, nbr, W, PW, NP, R,
                       E->order_cnt.p);
    if (exclusive_scan<int64_t>(E, E->order_cnt.p, nkeys, E->order_base.p, nullptr, false, st)) return -1;
    hipLaunchKernelGGL(order_len_cls, dim3(1), dim3(1), 0, st, E->order_base.p, c_nblks, NP, R, E->cls_lens.p);
    ACC_CHECK(hipMemcpyAsync(E->cls_host_lens, E->cls_lens.p, (2 * kNumClasses + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  } else {
  hipLaunchKernelGGL(order_count, grid_for(nkeys), dim3(256), 0, st, E->c_pre.p, E->row_nnz.p, nbr, W, PW, NP, NG, RG, E->order_cnt.p);
  if (exclusive_scan<int64_t>(E, E->order_cnt.p, nkeys, E->order_base.p, nullptr, false, st)) return -1;
  hipLaunchKernelGGL(order_len, dim3(1), dim3(1), 0, st, E->order_base.p, c_nblks, NP, NG, dsc + 7);
  }
  if (E->prod_cnt.ensure((size_t)c_nblks + 1) || E->blk_nze.ensure((size_t)c_nblks + 1) || E->prod_start.ensure((size_t)c_nblks + 1) ||
      E->c_blk_p_ws.ensure((size_t)c_nblks + 1))
    return -1;
  // 3. per C block: number of products, size; flop
  // one lane per (row, column) candidate unless C is extremely sparse (then one thread per bitmap word)
  const int nJ = (nbc + 63) / 64;
  E->grid_kernels = filtering || (((int64_t)nbr * nJ * 64 <= 256 * std::max<int64_t>(c_nblks, 1)) && !E->force_word_kernels);
  // sparse C (less than 60 % of the candidates are blocks) with enough block rows to fill the chip: product-driven kernels
  E->rows_kernels = E->force_symbolic == 3 || (E->force_symbolic == 0 && nbr >= 2048 && 10 * c_nblks < 6 * (int64_t)nbr * nbc);
  if (E->rows_kernels) {
    E->grid_kernels = false;
    ACC_CHECK(hipMemsetAsync(E->prod_cnt.p, 0, sizeof(int) * (size_t)c_nblks, st));
    hipLaunchKernelGGL(block_sizes_rows, grid_for((int64_t)nbr * W), dim3(256), 0, st, E->c_bm.p, E->c_pre.p, c_out_row_p, a->row_blk_size,
                       b->col_blk_size, nbr, W, E->blk_nze.p);
    hipLaunchKernelGGL(count_products_rows, grid_for((int64_t)nbr * 64), dim3(256), 0, st, a->row_p, a->col_i, a->row_blk_size, a->col_blk_size,
                       b->col_blk_size, b->row_p, b->col_i, E->c_bm.p, E->c_pre.p, c_out_row_p, nbr, W, E->prod_cnt.p, E->dev_scalars.p + 3,
                       E->filter);
  } else if (E->grid_kernels)
    hipLaunchKernelGGL(count_products_grid, grid_for((int64_t)nbr * nJ * 64), dim3(256), 0, st, a->row_p, a->col_i, a->row_blk_size,
                       a->col_blk_size, b->col_blk_size, E->b_bm.p, E->c_bm.p, E->c_pre.p, c_out_row_p, nbr, nbc, W, nJ,
                       E->prod_cnt.p, E->blk_nze.p, E->dev_scalars.p + 3, b->row_p, E->b_pre.p, E->filter);
  else
    hipLaunchKernelGGL(count_products, grid_for((int64_t)nbr * W), dim3(256), 0, st, a->row_p, a->col_i, a->row_blk_size,
                       a->col_blk_size, b->col_blk_size, E->b_bm.p, E->c_bm.p, E->c_pre.p, c_out_row_p, nbr, W, E->prod_cnt.p,
                       E->blk_nze.p, E->dev_scalars.p + 3);
  if (exclusive_scan<int64_t>(E, E->prod_cnt.p, c_nblks, E->prod_start.p, dsc + 2, false, st)) return -1;
  if (exclusive_scan<int64_t>(E, E->blk_nze.p, c_nblks, E->c_blk_p_ws.p, dsc + 1, false, st)) return -1;
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  E->order_len = E->host_scalars[7];
  if (E->cls_mode) {
    for (int c = 0; c < kNumClasses; ++c) {
      E->cls_len[c] = E->cls_host_lens[c];
      E->cls_off[c] = E->cls_host_lens[kNumClasses + c];
    }
    const int64_t total = E->cls_host_lens[2 * kNumClasses];
    E->order_len = total / 8;  // (only its product with 8 is used below: the size of order[])
    if (E->order.ensure((size_t)total + 64)) return -1;
    if (total > 0) {
      ACC_CHECK(hipMemsetAsync(E->order.p, 0xff, sizeof(int) * ((size_t)total + 64), st));
      hipLaunchKernelGGL(order_fill_cls, grid_for((int64_t)nbr * W), dim3(256), 0, st, E->c_bm.p, E->c_pre.p, c_out_row_p, E->cls_row.p,
                         E->cls_col_bm.p, E->order_base.p, E->cls_lens.p, nbr, W, PW, NP, R, E->order.p);
    }
  } else {
  if (E->order.ensure((size_t)(8 * E->order_len) + 64)) return -1;
  if (E->order_len > 0) {
    ACC_CHECK(hipMemsetAsync(E->order.p, 0xff, sizeof(int) * ((size_t)(8 * E->order_len) + 64), st));  // padding included
    hipLaunchKernelGGL(order_fill, grid_for((int64_t)nbr * W), dim3(256), 0, st, E->c_bm.p, E->c_pre.p, E->row_nnz.p, c_out_row_p,
                       E->order_base.p, nbr, W, PW, NP, NG, RG, E->order_len, E->order.p);
  }
  }
  counts->c_nblks = E->host_scalars[0];
  counts->c_nze = E->host_scalars[1];
  counts->nproducts = E->host_scalars[2];
  counts->flop = E->host_scalars[3];
  E->c_nblks = counts->c_nblks;
  E->nproducts = counts->nproducts;
  E->valid = true;
  ++E->plan_misses;
  if (!filtering) {
    if (plan_save(E, a, b, c_in, retain_sparsity ? 1 : 0, c_out_row_p, *counts, st)) return -1;
  } else {
    plan_invalidate(E);
  }
  return check(hipGetLastError(), "dbcsr_amd_mm_symbolic", __FILE__, __LINE__);
}

int dbcsr_amd_mm_numeric(void* handle, libsmm_acc_data_t datatype, double alpha, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b,
                         double beta, const dbcsr_amd_bcsr* c_in, dbcsr_amd_bcsr* c_out, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !E->valid || !a || !b || !c_in || !c_out) {
    fprintf(stderr, "dbcsr_amd_mm_numeric: no valid symbolic phase for this handle\n");
    return -1;
  }
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int nbr = E->nbr, W = E->W;
  const int64_t nblk = E->c_nblks;
  if (nblk == 0) return 0;
  // plan reuse: product lists, descriptors and launch order of the previous multiply stand; C's index is copied from the saved one
  const bool reuse = E->plan_hit && E->plan_numeric;
  if (!reuse) E->work_built = E->tile_built = E->band_built = E->group_built = false;
  if (E->entries.ensure((size_t)E->nproducts + 1) || E->descs.ensure((size_t)nblk + 1)) return -1;
  ACC_CHECK(hipEventRecord(E->ev[0], st));
  if (reuse) {
    ACC_CHECK(hipMemcpyAsync(c_out->col_i, E->plan_c_col_i.p, sizeof(int32_t) * (size_t)nblk, hipMemcpyDeviceToDevice, st));
    ACC_CHECK(hipMemcpyAsync(c_out->blk_p, E->plan_c_blk_p.p, sizeof(int64_t) * (size_t)nblk, hipMemcpyDeviceToDevice, st));
  } else if (E->rows_kernels) {
    if (E->tmp_i32.ensure((size_t)nblk + 1)) return -1;
    ACC_CHECK(hipMemsetAsync(E->tmp_i32.p, 0, sizeof(int) * (size_t)nblk, st));
    hipLaunchKernelGGL(fill_products_rows, grid_for((int64_t)nbr * 64), dim3(256), 0, st, a->row_p, a->col_i, a->blk_p, a->col_blk_size, b->row_p,
                       b->col_i, b->blk_p, E->c_bm.p, E->c_pre.p, c_out->row_p, E->prod_start.p, nbr, W, E->tmp_i32.p, E->entries.p, E->filter);
    hipLaunchKernelGGL(finish_descs_rows, grid_for((int64_t)nbr * W), dim3(256), 0, st, c_in->row_p, c_in->blk_p, c_out->row_blk_size,
                       c_out->col_blk_size, E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr,
                       E->have_cin ? E->cin_pre.p : (const int*)nullptr, E->c_bm.p, E->c_pre.p, c_out->row_p, E->c_blk_p_ws.p, E->prod_start.p,
                       E->prod_cnt.p, nbr, W, c_out->col_i, c_out->blk_p, E->descs.p);
  } else if (E->grid_kernels) {
    const int nbc = b->nblkcols, nJ = (nbc + 63) / 64;
    hipLaunchKernelGGL(fill_products_grid, grid_for((int64_t)nbr * nJ * 64), dim3(256), 0, st, a->row_p, a->col_i, a->blk_p, b->row_p,
                       b->blk_p, c_in->row_p, c_in->blk_p, a->row_blk_size, a->col_blk_size, b->col_blk_size, E->b_bm.p, E->b_pre.p,
                       E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr, E->have_cin ? E->cin_pre.p : (const int*)nullptr,
                       E->c_bm.p, E->c_pre.p, c_out->row_p, E->prod_start.p, E->c_blk_p_ws.p, nbr, nbc, W, nJ, c_out->col_i,
                       c_out->blk_p, E->descs.p, E->entries.p, E->filter);
  } else {
    hipLaunchKernelGGL(fill_products, grid_for((int64_t)nbr * W), dim3(256), 0, st, a->row_p, a->col_i, a->blk_p, b->row_p, b->blk_p,
                       c_in->row_p, c_in->blk_p, a->row_blk_size, a->col_blk_size, b->col_blk_size, E->b_bm.p, E->b_pre.p,
                       E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr, E->have_cin ? E->cin_pre.p : (const int*)nullptr,
                       E->c_bm.p, E->c_pre.p, c_out->row_p, E->prod_start.p, E->c_blk_p_ws.p, nbr, W, c_out->col_i, c_out->blk_p,
                       E->descs.p, E->entries.p);
  }
  const unsigned nwg = (unsigned)((nblk + 3) / 4);
  // waves per workgroup of the one-wave-per-C-block kernels.  A workgroup's LDS is released when its LAST wave ends: with short,
  // uneven product lists one wave per workgroup keeps more wave slots busy (config 3: kernel 8.93 -> 7.51 ms, generic LDS kernel
  // 12.3 -> 9.2, config 2: -5 %, config 4: -6 %); with long lists four waves per workgroup are faster (config 5, 164 products per
  // block: 2.03 s against 2.27 s)
  const int ww = E->wg_waves > 0 ? E->wg_waves : (E->nproducts <= 32 * nblk ? 1 : 4);
  // in-place accumulation (Cannon ticks after the first): C blocks without products in this call are left untouched
  const int skip_empty = (c_out->data == c_in->data && E->retain && beta == 1.0) ? 1 : 0;
  // launch-order work records for the exact-size fp64 kernels (one wave per C block): descriptor + first product in one read
  // the one-wave slab kernel (mm_numeric_f64_mid.h): fp64 C blocks whose dominant (else largest) size has a dimension of 33 ... 40 and the other of
  // 21 ... 40 (mid_f64_serves), any inner dimension; its second launch takes the blocks of another size.  Mixed-size multiplies (cls_mode) ask per class below.
  int mid_rb = 0, mid_cb = 0;
  if (datatype == dbcsr_type_real_8 && E->use_big && E->use_mid && E->use_lds && !E->cls_mode && E->max_m <= 40 && E->max_n <= 40 && E->min_m >= 1 &&
      E->min_n >= 1 && E->min_k >= 1 && E->order_len > 0 && !(E->dbg & ~32) && !E->dma_stages && !E->hot_persistent && E->hot_variant == 0 && E->use_hot &&
      E->use_pipe != 1) {
    // (without a dominant size -- the size statistics stop at 32 -- the largest size is multiplied exactly when the blocks go beyond 32, where the
    // alternative is the workgroup kernel, or when every block is in the range: the second launch pads the others to 40 x 40)
    const bool dom = E->hot_m > 0 && E->hot_n > 0, all_in = (E->min_m > 24 && E->min_n > 24) || E->max_m > 32 || E->max_n > 32;
    const int ur = dom ? (E->hot_m + 3) / 4 : (all_in ? (E->max_m + 3) / 4 : 0), uc = dom ? (E->hot_n + 3) / 4 : (all_in ? (E->max_n + 3) / 4 : 0);
    if (mid_f64_serves(ur, uc, 0)) mid_rb = ur, mid_cb = uc;
  }
  const Work* hot_work = nullptr;
  {
    const bool small64 = datatype == dbcsr_type_real_8 && E->use_lds && E->max_m <= 32 && E->max_k <= 32 && E->max_n <= 32 && E->min_m >= 1 &&
                         E->min_k >= 1 && E->min_n >= 1 && !(E->use_tiny && E->max_m <= 4 && E->max_n <= 4);
    // (the ahead-of-time exact-size kernel reads nothing else; the class kernels keep the order[] -> descs[] path for DBCSR_AMD_MM_WORK=0)
    const bool exact = E->cls_mode ? (E->class_g == 1 && E->use_work)
                                   : (E->use_hot && E->use_pipe != 1 && E->hot_m > 0 && E->dma_stages == 0 && E->hot_m == E->hot_n && E->hot_m == E->hot_k);
    const int64_t npos = 8 * E->order_len;
    if (((small64 && exact) || mid_rb) && npos > 0) {
      if (!(reuse && E->work_built)) {
        if (E->work.ensure((size_t)npos + 1)) return -1;
        hipLaunchKernelGGL(build_work, grid_for(npos), dim3(256), 0, st, E->order.p, npos, E->descs.p, nblk, E->entries.p, E->work.p);
        E->work_built = true;
      }
      hot_work = E->work.p;
    }
  }
  // a filtered multiply ends with the block filter on C's norms: the exact-size kernel leaves them behind (dbcsr_amd_bcsr_filter_count
  // then skips its pass over C)
  double* epi_norms = nullptr;
  E->norms_data = nullptr;
  if ((hot_work || (E->cls_mode && E->class_g == 1)) && datatype == dbcsr_type_real_8 && E->filter.a_norms && !skip_empty && !E->retain) {
    if (E->norms64.ensure((size_t)nblk + 1)) return -1;
    epi_norms = E->norms64.p;
    if (E->dbg & 8) epi_norms = nullptr;  // (profiling epilogue of the exact-size kernel: it leaves no norms, the filter then computes them)
  }
  ACC_CHECK(hipEventRecord(E->ev[1], st));
  if (datatype == dbcsr_type_real_8) {
    // LDS path: blocks of at most 32 x 32 (any smaller size: the staging loads are bounds-checked buffer loads)
    const bool small = E->max_m <= 32 && E->max_k <= 32 && E->max_n <= 32 && E->min_m >= 1 && E->min_k >= 1 && E->min_n >= 1;
    if (E->use_tiny && E->max_m <= 4 && E->max_n <= 4 && E->min_m >= 1 && E->min_n >= 1 && E->min_k >= 1) {
      // four C blocks per wave, one per MFMA sub-block; order[] is padded to a multiple of 4 per XCD stream, so a wave's
      // four positions never straddle two streams only if the stream length is a multiple of 16: the tail positions hold -1
      const unsigned nwg_t = (unsigned)((8 * E->order_len + 15) / 16);
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_tiny");
      const double* ad = static_cast<const double*>(a->data);
      const double* bd = static_cast<const double*>(b->data);
      double* cd = static_cast<double*>(c_out->data);
      const double* cid = static_cast<const double*>(c_in->data);
      if (nwg_t > 0) {
        auto tiny = E->max_k > 4 ? mm_numeric_f64_tiny<false> : mm_numeric_f64_tiny<true>;
        hipLaunchKernelGGL(tiny, dim3(nwg_t), dim3(256), 0, st, E->descs.p, nblk, E->entries.p, ad, bd, cd, cid, alpha, beta, skip_empty, E->order.p);
      }
    } else if (mid_rb && launch_mid_f64(mid_rb, mid_cb, E->min_m != E->max_m || E->min_n != E->max_n, (unsigned)(8 * E->order_len), st, E->descs.p, nblk,
                                         E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                                         static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, skip_empty, E->order.p,
                                         hot_work)) {
      // blocks of 25 ... 40 in both dimensions: one wave per C block, operands in slabs (mm_numeric_f64_mid.h); the dominant size (else the largest)
      // multiplied exactly, the blocks of another size by the second launch.  (It leaves no block norms: a filtered multiply computes them.)
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_mid<%d,%d>", mid_rb, mid_cb);
    } else if (small && E->use_lds && E->cls_mode) {
      // one launch per (m, n) class on its segment of order[]: the run-time compiled exact-size kernel of the class
      // (mm_exact.h, mm_jit.hip), the generic LDS kernel for class 9 (other sizes) and for classes hiprtc could not serve
      const int g_lds_a = (E->max_m * ((E->max_k + 3) & ~3) + 1) & ~1, g_lds_b = ((E->max_k * E->max_n + 127) / 128) * 128;
      const int g_lds_wave = g_lds_a + g_lds_b;
      const int g_maxt = (std::max(E->max_m, E->max_n) + 7) / 8;
      const int dbgv = E->dbg | (skip_empty ? 32 : 0);
      int njit = 0, ngen = 0, nmid = 0, jit_mask = 0;
      const hipStream_t st_main = st;
      int nside = 0, nlaunch = 0;
      if (E->class_streams > 1) {
        nside = E->class_streams - 1;
        if (!E->fork_ev) ACC_CHECK(hipEventCreateWithFlags(&E->fork_ev, hipEventDisableTiming));
        ACC_CHECK(hipEventRecord(E->fork_ev, st_main));
        for (int i = 0; i < nside; ++i) {
          if (!E->side_stream[i]) {
            ACC_CHECK(hipStreamCreateWithFlags(&E->side_stream[i], hipStreamNonBlocking));
            ACC_CHECK(hipEventCreateWithFlags(&E->join_ev[i], hipEventDisableTiming));
          }
          ACC_CHECK(hipStreamWaitEvent(E->side_stream[i], E->fork_ev, 0));
        }
      }
      for (int c = 0; c < kNumClasses; ++c) {
        if (E->cls_len[c] == 0) continue;
        {
          const int slot = nlaunch++ % (nside + 1);
          st = slot == 0 ? st_main : E->side_stream[slot - 1];
        }
        const int* ord = E->order.p + E->cls_off[c];
        const unsigned nwg_c = (unsigned)(8 * E->cls_len[c] / 4);
        ClassKernel ck;
        const int cm = c < 9 ? E->cls_m[c / 3] : 0, cn = c < 9 ? E->cls_n[c % 3] : 0;
        // classes of 29 ... 32 rows and columns, or 21 ... 24 in one of them (DBCSR_AMD_MM_MID=3: not those): the one-wave slab kernel -- half the LDS of
        // the class kernel, which stages whole blocks (17.9 KB per wave for (32, 32), 15 KB for (32, 23): two waves per SIMD)
        if (c < 9 && E->use_mid && E->use_big && E->class_g == 1 && mid_f64_serves((cm + 3) / 4, (cn + 3) / 4, E->use_mid == 3 ? 3 : 1) &&
            launch_mid_f64((cm + 3) / 4, (cn + 3) / 4, false, (unsigned)(8 * E->cls_len[c]), st, E->descs.p, nblk, E->entries.p,
                           static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),
                           static_cast<const double*>(c_in->data), alpha, beta, skip_empty, ord, hot_work ? hot_work + E->cls_off[c] : nullptr)) {
          ++nmid;
        } else if (c < 9 && cm > 0 && cn > 0 && jit_class_kernel(cm, cn, E->cls_k[0], E->cls_k[1], E->cls_k[2], E->class_g, &ck) == 0) {
          const Desc* p_descs = E->descs.p;
          long p_nblk = (long)nblk;
          const Entry* p_entries = E->entries.p;
          const double* p_a = static_cast<const double*>(a->data);
          const double* p_b = static_cast<const double*>(b->data);
          double* p_c = static_cast<double*>(c_out->data);
          const double* p_ci = static_cast<const double*>(c_in->data);
          double p_alpha = alpha, p_beta = beta;
          int p_skip = skip_empty;
          const Work* p_work = hot_work ? hot_work + E->cls_off[c] : nullptr;
          double* p_norms = epi_norms;
          void* args[] = {&p_descs, &p_nblk, &p_entries, &p_a, &p_b, &p_c, &p_ci, &p_alpha, &p_beta, &p_skip, &ord, &p_work, &p_norms};
          jit_mask |= 1 << c;
          const unsigned cw = E->class_g == 1 ? (unsigned)ww : 4u;  // waves per workgroup (the G-block stream body keeps 4)
          ACC_CHECK(hipModuleLaunchKernel(ck.fn, (unsigned)(8 * E->cls_len[c]) / cw / (unsigned)E->class_g, 1, 1, 64 * cw, 1, 1,
                                          (unsigned)(cw * ck.wave_lds), st, args, nullptr));
          ++njit;
        } else {
          const size_t lb = (size_t)ww * g_lds_wave * sizeof(double);
#define DBCSR_LAUNCH_G(T_)                                                                                                         \
  hipLaunchKernelGGL(mm_numeric_f64_lds<T_>, dim3(nwg_c * 4u / (unsigned)ww), dim3(64 * ww), lb, st, E->descs.p, nblk, E->entries.p, \
                     static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),   \
                     static_cast<const double*>(c_in->data), alpha, beta, g_lds_a, g_lds_wave, dbgv, ord)
          switch (g_maxt) {
            case 1: DBCSR_LAUNCH_G(1); break;
            case 2: DBCSR_LAUNCH_G(2); break;
            case 3: DBCSR_LAUNCH_G(3); break;
            default: DBCSR_LAUNCH_G(4); break;
          }
#undef DBCSR_LAUNCH_G
          ++ngen;
        }
      }
      st = st_main;
      for (int i = 0; i < nside; ++i) {
        ACC_CHECK(hipEventRecord(E->join_ev[i], E->side_stream[i]));
        ACC_CHECK(hipStreamWaitEvent(st_main, E->join_ev[i], 0));
      }
      if (epi_norms) {  // the blocks the generic kernel handled did not leave their norm
        ClassSet cs;
        for (int q = 0; q < 3; ++q) cs.m[q] = E->cls_m[q], cs.n[q] = E->cls_n[q];
        cs.jit_mask = jit_mask;
        hipLaunchKernelGGL(block_norms_unserved_classes, grid_for(nblk * 64), dim3(256), 0, st, E->descs.p, nblk,
                           static_cast<const double*>(c_out->data), cs, epi_norms);
        E->norms_data = c_out->data;
        E->norms_nblks = nblk;
      }
      if (nmid > 0)
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_class[%d jit + %d slab + %d generic launches; m {%d,%d,%d} n {%d,%d,%d} k {%d,%d,%d}]",
                 njit, nmid, ngen, E->cls_m[0], E->cls_m[1], E->cls_m[2], E->cls_n[0], E->cls_n[1], E->cls_n[2], E->cls_k[0], E->cls_k[1], E->cls_k[2]);
      else
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_class[%d jit + %d generic launches; m {%d,%d,%d} n {%d,%d,%d} k {%d,%d,%d}]", njit,
               ngen, E->cls_m[0], E->cls_m[1], E->cls_m[2], E->cls_n[0], E->cls_n[1], E->cls_n[2], E->cls_k[0], E->cls_k[1], E->cls_k[2]);
    } else if (small && E->use_lds) {
      // per-wave LDS slice.  Staging writes whole 1 KiB chunks (128 doubles), A's chunks first, then B's: the B part may
      // start right after A's (zero-padded) block -- the tail of A's last chunk is simply overwritten by B's first chunk
      // (one wave, in-order LDS queue) -- and only B's part is rounded up to whole chunks.  For 23x23 blocks this is
      // 9.5 KB per wave instead of 10 KB, which is what lets a 4th workgroup (16 waves) fit the CU's 160 KB.
      int lds_a = (E->max_m * ((E->max_k + 3) & ~3) + 1) & ~1, lds_b = ((E->max_k * E->max_n + 127) / 128) * 128;
      if (E->hot_m > 0 && E->hot_m == E->hot_n && E->hot_m == E->hot_k && E->hot_m % 16 == 0) {
        // the exact-size kernel stages columns of 16 / 32 doubles with a pitch of + 2 (mm_numeric_f64.h: cblock_f64_exact): its A image has hot_m + 2
        // rows per column, its B image 16 more bytes per column (128 per KiB piece at most)
        const int S = E->hot_m, cb = (S * S * 8 + 1023) / 1024;
        lds_a = std::max(lds_a, (S + 2) * S);
        lds_b = std::max(lds_b, cb * (1024 + 128) / 8 + 2);
      }
      const int lds_wave = lds_a + lds_b;
      const int maxt = (std::max(E->max_m, E->max_n) + 7) / 8;
      const size_t lds_bytes = (size_t)4 * lds_wave * sizeof(double) + (size_t)E->lds_pad;
      const unsigned nwg_o = (unsigned)(8 * E->order_len / 4);
#define DBCSR_LAUNCH(T_)                                                                                                        \
  hipLaunchKernelGGL(mm_numeric_f64_lds<T_>, dim3(nwg_o * 4u / (unsigned)ww), dim3(64 * ww),                     \
                     (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad, st, E->descs.p, nblk, E->entries.p,     \
                     static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data), \
                     static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, E->dbg | (skip_empty ? 32 : 0), E->order.p)
      int tile_rc = 1;
#ifdef DBCSR_AMD_EXPERIMENTS
      // a wave per R C blocks of one block column, B shared inside the wave (mm_group64.h): one dominant cube size the kernel is built for, no
      // block norms to leave behind (filtered multiplies), no symmetric product
      if (E->f64_group >= 2 && hot_work && E->use_hot && E->use_pipe != 1 && E->dma_stages == 0 && E->hot_m == E->hot_n && E->hot_m == E->hot_k && !epi_norms &&
          !E->canonical_c && !(E->dbg & ~32) && !E->hot_persistent) {
        tile_rc = run_group_f64(E, E->f64_group, reuse, st, a, b, c_in, c_out, alpha, beta, skip_empty);
        if (tile_rc < 0) return -1;
        if (tile_rc == 0) tile_rc = 3;
      }
      // XCD-wide C tiles (mm_tile.h): one dominant cube size the tile kernel is built for, a C dense enough that sub-tiles of
      // 3 x 3 blocks have long product lists, no on-the-fly filter, no in-place accumulation, no symmetric product
      if (tile_rc != 3 && E->use_tile > 0 && hot_work && E->use_hot && E->use_pipe != 1 && E->dma_stages == 0 && E->hot_m == 23 && E->hot_n == 23 && E->hot_k == 23 &&
          !E->filter.a_norms && !skip_empty && !E->canonical_c && !epi_norms && !(E->dbg & ~32) &&
          (E->use_tile > 1 || (E->nproducts >= 8 * nblk && nblk >= 200000)))
        tile_rc = run_tile_f64<23>(E, st, a, b, c_in, c_out, alpha, beta);
      if (tile_rc < 0) return -1;
      // CU-wide C tiles, B shared in LDS (mm_band.h): the same conditions, and no retain_sparsity (its lists take C's pattern from the operands)
      if (tile_rc != 0 && tile_rc != 3 && E->use_band > 0 && hot_work && E->use_hot && E->use_pipe != 1 && E->dma_stages == 0 && E->hot_m == 23 && E->hot_n == 23 &&
          E->hot_k == 23 && !E->filter.a_norms && !skip_empty && !E->canonical_c && !epi_norms && !(E->dbg & ~32) && !E->retain && !E->hot_persistent &&
          (E->use_band > 1 || (E->nproducts >= 8 * nblk && nblk >= 200000))) {
        tile_rc = run_band_f64<23>(E, st, a, b, c_in, c_out, alpha, beta);
        if (tile_rc < 0) return -1;
        if (tile_rc == 0) tile_rc = 2;
      }
      // measured: the pipelined kernel wins when C blocks have few products (config 3: 3.7 per block, 10.4 vs 11.8 ms) and
      // loses when they have many (config 2: 14.4 per block, 32 vs 22 ms)
#endif
      if (tile_rc == 3) {
        // the C blocks of other sizes (tail block row / column): the one-wave-per-block kernel, told to leave the dominant size alone
        if (E->hot_cnt_m < nbr || E->hot_cnt_n < b->nblkcols)
          launch_hot_f64(E->hot_m, E->hot_n, E->hot_k, dim3((unsigned)(8 * E->order_len / ww)), (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad,
                         st, E->descs.p, nblk, E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                         static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, 64 | (skip_empty ? 32 : 0),
                         E->order.p, hot_work, ww, nullptr, 0);
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_group<%d,%d,%d;%d>", E->hot_m, E->hot_n, E->hot_k, E->group_R);
      } else if (tile_rc == 0 || tile_rc == 2) {
        // the tile / band kernel computed the C blocks of the dominant size (products with inner blocks of another size included);
        // this launch: the exact-size kernel over the blocks of the other sizes only
        launch_hot_f64(E->hot_m, E->hot_n, E->hot_k, dim3((unsigned)(8 * E->order_len / ww)), (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad,
                       st, E->descs.p, nblk, E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                       static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, 64, E->order.p,
                       hot_work, ww, nullptr, 0);
        snprintf(E->last_kernel, sizeof E->last_kernel, tile_rc == 2 ? "mm_numeric_f64_band<%d,%d,%d>" : "mm_numeric_f64_tile<%d,%d,%d>", E->hot_m, E->hot_n,
                 E->hot_k);
#ifdef DBCSR_AMD_EXPERIMENTS
      } else if (E->use_hot && E->use_pipe != 1 && E->hot_m > 0 && E->dma_stages > 0 &&
          launch_dma_f64(E->dma_stages, E->hot_m, E->hot_n, E->hot_k, (unsigned)(8 * E->order_len), st, E->descs.p, nblk, E->entries.p,
                         static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),
                         static_cast<const double*>(c_in->data), alpha, beta, skip_empty, E->order.p)) {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_dma<%d,%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k, E->dma_stages);
      } else if (E->hot_persistent && E->use_hot && E->use_pipe != 1 && E->hot_m == 23 && E->hot_n == 23 && E->hot_k == 23 && hot_work &&
                 !(E->dbg & ~32) && E->hot_counters.ensure(8 * 32) == 0) {
        // persistent waves, one counter per XCD (an experiment: see the kernel); 16 one-wave workgroups per CU is what the LDS slice allows
        static int n_cu_p = 0;
        if (n_cu_p == 0) {
          int dev = 0;
          ACC_CHECK(hipGetDevice(&dev));
          ACC_CHECK(hipDeviceGetAttribute(&n_cu_p, hipDeviceAttributeMultiprocessorCount, dev));
        }
        const int per_cu = std::max(1, (int)((160 * 1024) / ((size_t)lds_wave * sizeof(double) + (size_t)E->lds_pad)));
        ACC_CHECK(hipMemsetAsync(E->hot_counters.p, 0, 8 * 32 * sizeof(unsigned), st));
        hipLaunchKernelGGL((mm_numeric_f64_hot_persistent<23, 23, 23>), dim3((unsigned)(n_cu_p * per_cu)), dim3(64),
                           (size_t)lds_wave * sizeof(double) + (size_t)E->lds_pad, st, E->entries.p, static_cast<const double*>(a->data),
                           static_cast<const double*>(b->data), static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta,
                           lds_a, skip_empty ? 32 : 0, hot_work, (long)E->order_len, E->hot_counters.p, E->hot_xcd_mask, epi_norms);
        if (epi_norms) {
          hipLaunchKernelGGL(block_norms_other_sizes, grid_for(nblk * 64), dim3(256), 0, st, E->descs.p, nblk, static_cast<const double*>(c_out->data),
                             E->hot_m, E->hot_n, epi_norms);
          E->norms_data = c_out->data;
          E->norms_nblks = nblk;
        }
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_hot_persistent<%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k);
#endif
      } else if (E->use_hot && E->use_pipe != 1 && E->hot_m > 0 &&
          launch_hot_f64(E->hot_m, E->hot_n, E->hot_k, dim3((unsigned)(8 * E->order_len / ww)),
                         (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad, st, E->descs.p, nblk, E->entries.p,
                         static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),
                         static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, E->dbg | (skip_empty ? 32 : 0), E->order.p,
                         hot_work, ww, epi_norms, (E->dbg & ~32) ? 1 : E->hot_variant)) {
        // launched: C blocks of the dominant size take the exact-size path, the others the generic one
        if (epi_norms) {  // blocks of another size (tail block row / column) did not leave their norm: a pass over those only
          hipLaunchKernelGGL(block_norms_other_sizes, grid_for(nblk * 64), dim3(256), 0, st, E->descs.p, nblk, static_cast<const double*>(c_out->data),
                             E->hot_m, E->hot_n, epi_norms);
          E->norms_data = c_out->data;
          E->norms_nblks = nblk;
        }
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_hot<%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k);
      } else {
      const bool pipe = E->use_pipe == 1 || (E->use_pipe < 0 && E->nproducts < 6 * nblk && E->nproducts > nblk + nblk / 2);
      if (pipe) {
        const int64_t npos = 8 * E->order_len;
        const int G = E->pipe_g;
        const unsigned nwg_p = (unsigned)((npos + 4 * (int64_t)G - 1) / (4 * (int64_t)G));
#define DBCSR_LAUNCH_P(T_)                                                                                                       \
  hipLaunchKernelGGL(mm_numeric_f64_pipe<T_>, dim3(nwg_p), dim3(256), lds_bytes, st, E->descs.p, nblk, E->entries.p,             \
                     static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data), \
                     static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, skip_empty, E->order.p, npos, G)
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_pipe<%d>", maxt > 4 ? 4 : maxt);
        switch (maxt) {
          case 1: DBCSR_LAUNCH_P(1); break;
          case 2: DBCSR_LAUNCH_P(2); break;
          case 3: DBCSR_LAUNCH_P(3); break;
          default: DBCSR_LAUNCH_P(4); break;
        }
#undef DBCSR_LAUNCH_P
      } else {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_lds<%d>", maxt > 4 ? 4 : maxt);
        switch (maxt) {
          case 1: DBCSR_LAUNCH(1); break;
          case 2: DBCSR_LAUNCH(2); break;
          case 3: DBCSR_LAUNCH(3); break;
          default: DBCSR_LAUNCH(4); break;
        }
      }
      }
#undef DBCSR_LAUNCH
    } else if (E->use_big && E->use_lds && E->max_m <= 80 && E->max_n <= 80 && E->min_m >= 1 && E->min_n >= 1 && E->min_k >= 1 && !E->cls_mode &&
               E->order_len > 0 && (E->max_m > 32 || E->max_n > 32 || ((E->max_m + 7) / 8) * ((E->max_n + 7) / 8) >= 4) &&
               launch_big_f64(std::max(2, ((E->max_m + 7) / 8 + 1) / 2), std::max(2, ((E->max_n + 7) / 8 + 1) / 2), (unsigned)(8 * E->order_len), st,
                              E->descs.p, nblk, E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                              static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta,
                              skip_empty | (E->use_big == 2 ? 4 : 0), E->order.p)) {
      // blocks of 33 ... 80 (or an inner dimension above 32): one workgroup per C block, operand slabs shared through LDS (mm_numeric_f64_big.h)
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_big<%d,%d>", std::max(2, ((E->max_m + 7) / 8 + 1) / 2),
               std::max(2, ((E->max_n + 7) / 8 + 1) / 2));
    } else {
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64");
      hipLaunchKernelGGL(mm_numeric_f64, dim3(nwg), dim3(256), 0, st, E->descs.p, nblk, E->entries.p,
                         static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                         static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, skip_empty);
    }
  } else {
    const bool small32 = E->max_m <= 32 && E->max_k <= 32 && E->max_n <= 32 && E->min_m >= 1 && E->min_k >= 1 && E->min_n >= 1;
    if (small32 && E->use_lds && E->cls_mode) {
      for (int c = 0; c < kNumClasses; ++c) {
        if (E->cls_len[c] == 0) continue;
        hipLaunchKernelGGL(mm_numeric_f32_lds, dim3((unsigned)(8 * E->cls_len[c] / ww)), dim3(64 * ww), f32_lds_bytes(ww), st, E->descs.p, nblk, E->entries.p,
                           static_cast<const float*>(a->data), static_cast<const float*>(b->data), static_cast<float*>(c_out->data),
                           static_cast<const float*>(c_in->data), (float)alpha, (float)beta, skip_empty, E->order.p + E->cls_off[c]);
      }
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32_lds[per class segment]");
    } else if (small32 && E->use_lds) {
      const unsigned nwg_o = (unsigned)(8 * E->order_len / ww);
      int grp_rc = 1;
#ifdef DBCSR_AMD_EXPERIMENTS
      if (E->use_hot && E->hot_m > 0 && E->hot_m == E->hot_n && E->hot_m == E->hot_k && E->f32_direct && E->f32_group != 0 && E->min_k == E->max_k &&
          E->max_k == E->hot_k && (E->f32_group > 0 || (E->nproducts >= 16 * nblk && nblk >= 1024))) {
        grp_rc = run_group_f32(E, E->f32_group > 0 ? E->f32_group : 4, reuse, st, a, b, c_in, c_out, (float)alpha, (float)beta, skip_empty);
        if (grp_rc < 0) return -1;
      }
#endif
      if (grp_rc == 0) {
        // the C blocks of other sizes (tail block row / column): the one-wave-per-block kernel, told to leave the dominant size alone
        if (E->hot_cnt_m < nbr || E->hot_cnt_n < b->nblkcols)
          launch_hot_f32_direct(E->hot_m, E->hot_n, E->hot_k, dim3(nwg_o), ww, st, E->descs.p, nblk, E->entries.p, static_cast<const float*>(a->data),
                                static_cast<const float*>(b->data), static_cast<float*>(c_out->data), static_cast<const float*>(c_in->data),
                                (float)alpha, (float)beta, skip_empty | 2, E->order.p);
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32_group<%d,%d,%d;%d>", E->hot_m, E->hot_n, E->hot_k, E->group_R);
      } else if (E->use_hot && E->hot_m > 0 && E->f32_direct &&
          launch_hot_f32_direct(E->hot_m, E->hot_n, E->hot_k, dim3(nwg_o), ww, st, E->descs.p, nblk, E->entries.p, static_cast<const float*>(a->data),
                                static_cast<const float*>(b->data), static_cast<float*>(c_out->data), static_cast<const float*>(c_in->data),
                                (float)alpha, (float)beta, skip_empty, E->order.p,
                                E->f32_direct >= 2 && E->hot_cnt_m == nbr && E->hot_cnt_n == b->nblkcols)) {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32_direct<%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k);
      } else if (E->use_hot && E->hot_m > 0 &&
          launch_hot_f32(E->hot_m, E->hot_n, E->hot_k, dim3(nwg_o), ww, st, E->descs.p, nblk, E->entries.p, static_cast<const float*>(a->data),
                         static_cast<const float*>(b->data), static_cast<float*>(c_out->data), static_cast<const float*>(c_in->data),
                         (float)alpha, (float)beta, skip_empty, E->order.p)) {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32_hot<%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k);
      } else {
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32_lds");
      hipLaunchKernelGGL(mm_numeric_f32_lds, dim3(nwg_o), dim3(64 * ww), f32_lds_bytes(ww), st, E->descs.p, nblk, E->entries.p,
                         static_cast<const float*>(a->data), static_cast<const float*>(b->data), static_cast<float*>(c_out->data),
                         static_cast<const float*>(c_in->data), (float)alpha, (float)beta, skip_empty, E->order.p);
      }
    } else {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32");
        hipLaunchKernelGGL(mm_numeric_f32, dim3(nwg), dim3(256), 0, st, E->descs.p, nblk, E->entries.p,
                       static_cast<const float*>(a->data), static_cast<const float*>(b->data), static_cast<float*>(c_out->data),
                       static_cast<const float*>(c_in->data), (float)alpha, (float)beta, skip_empty);
    }
  }
  ACC_CHECK(hipEventRecord(E->ev[2], st));
  E->timed = true;
  c_out->nblks = nblk;
  if (E->plan_saved && !E->plan_numeric) {  // first numeric phase of a saved plan: keep C's index for the multiplies that reuse it
    if (E->plan_c_col_i.ensure((size_t)nblk + 1) || E->plan_c_blk_p.ensure((size_t)nblk + 1)) return -1;
    ACC_CHECK(hipMemcpyAsync(E->plan_c_col_i.p, c_out->col_i, sizeof(int32_t) * (size_t)nblk, hipMemcpyDeviceToDevice, st));
    ACC_CHECK(hipMemcpyAsync(E->plan_c_blk_p.p, c_out->blk_p, sizeof(int64_t) * (size_t)nblk, hipMemcpyDeviceToDevice, st));
    E->plan_numeric = true;
  }
  return check(hipGetLastError(), "dbcsr_amd_mm_numeric", __FILE__, __LINE__);
}


int dbcsr_amd_mm_init_c(void* handle, libsmm_acc_data_t datatype, double beta, const dbcsr_amd_bcsr* c_in, dbcsr_amd_bcsr* c_out,
                        void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) E->plan_numeric = false;  // the descriptors are rewritten below: a numeric phase that follows fills its lists again (the plan itself stands)
  if (!E || !E->valid || !c_in || !c_out) {
    fprintf(stderr, "dbcsr_amd_mm_init_c: no valid symbolic phase for this handle\n");
    return -1;
  }
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int nbr = E->nbr, W = E->W;
  const int64_t nblk = E->c_nblks;
  c_out->nblks = nblk;
  if (nblk == 0) return 0;
  if (E->descs.ensure((size_t)nblk + 1)) return -1;
  hipLaunchKernelGGL(emit_index, grid_for((int64_t)nbr * W), dim3(256), 0, st, c_in->row_p, c_in->blk_p, c_out->row_blk_size,
                     c_out->col_blk_size, E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr,
                     E->have_cin ? E->cin_pre.p : (const int*)nullptr, E->c_bm.p, E->c_pre.p, c_out->row_p, E->c_blk_p_ws.p, nbr, W,
                     c_out->col_i, c_out->blk_p, E->descs.p);
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((init_c_blocks<double>), grid_for(nblk * 64), dim3(256), 0, st, E->descs.p, nblk,
                       static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), beta);
  else
    hipLaunchKernelGGL((init_c_blocks<float>), grid_for(nblk * 64), dim3(256), 0, st, E->descs.p, nblk,
                       static_cast<float*>(c_out->data), static_cast<const float*>(c_in->data), (float)beta);
  return check(hipGetLastError(), "dbcsr_amd_mm_init_c", __FILE__, __LINE__);
}


static int element_offsets(Engine* E, const int* sizes, int n, DevBuf<int64_t>& off, hipStream_t st) {
  if (off.ensure((size_t)n + 1)) return -1;
  return exclusive_scan<int64_t>(E, sizes, n, off.p, nullptr, false, st);
}

static Window make_window(const dbcsr_amd_bcsr* m, int64_t row_lo, int64_t row_hi, int64_t col_lo, int64_t col_hi) {
  (void)m;
  const int64_t big = 0x7fffffff;
  Window w;
  w.r0 = (int)(row_lo < 0 ? 0 : row_lo);
  w.r1 = (int)(row_hi < 0 || row_hi > big ? big : row_hi);
  w.c0 = (int)(col_lo < 0 ? 0 : col_lo);
  w.c1 = (int)(col_hi < 0 || col_hi > big ? big : col_hi);
  return w;
}

int dbcsr_amd_bcsr_crop_count(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, int64_t row_lo, int64_t row_hi,
                              int64_t col_lo, int64_t col_hi, int32_t* new_row_p, int64_t* new_nblks, int64_t* new_nze, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !m || !new_row_p || !new_nblks || !new_nze) return -1;
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int nbr = m->nblkrows;
  const int64_t nb = m->nblks;
  E->valid = false;  // shares workspace with the symbolic phase
  E->flt_nblks = nb;
  E->crop_win = make_window(m, row_lo, row_hi, col_lo, col_hi);
  E->crop_pending = true;
  if (E->keep.ensure((size_t)nb + 1) || E->blk_nze.ensure((size_t)nb + 1) || E->row_nnz.ensure((size_t)nbr + 1) ||
      E->prod_start.ensure((size_t)nb + 1) || E->c_blk_p_ws.ensure((size_t)nb + 1) || E->dev_scalars.ensure(16))
    return -1;
  int64_t* dsc = reinterpret_cast<int64_t*>(E->dev_scalars.p);
  ACC_CHECK(hipMemsetAsync(dsc, 0, 16 * sizeof(int64_t), st));
  if (element_offsets(E, m->row_blk_size, nbr, E->off_a, st)) return -1;
  if (element_offsets(E, m->col_blk_size, m->nblkcols, E->off_b, st)) return -1;
  if (nbr > 0 && nb > 0)
    hipLaunchKernelGGL(crop_flags, grid_for((int64_t)nbr * 64), dim3(256), 0, st, m->row_p, m->col_i, m->row_blk_size, m->col_blk_size,
                       E->off_a.p, E->off_b.p, nbr, E->crop_win, E->keep.p, E->blk_nze.p, E->row_nnz.p);
  else if (nbr > 0)
    ACC_CHECK(hipMemsetAsync(E->row_nnz.p, 0, sizeof(int) * (size_t)nbr, st));
  if (exclusive_scan<int32_t>(E, E->row_nnz.p, nbr, new_row_p, dsc + 0, true, st)) return -1;
  if (exclusive_scan<int64_t>(E, E->keep.p, nb, E->prod_start.p, nullptr, false, st)) return -1;
  if (exclusive_scan<int64_t>(E, E->blk_nze.p, nb, E->c_blk_p_ws.p, dsc + 1, false, st)) return -1;
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  *new_nblks = E->host_scalars[0];
  *new_nze = E->host_scalars[1];
  return check(hipGetLastError(), "dbcsr_amd_bcsr_crop_count", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_crop_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, dbcsr_amd_bcsr* dst, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !src || !dst || !E->crop_pending || E->flt_nblks != src->nblks) return -1;
  E->crop_pending = false;
  hipStream_t st = stream_of(stream);
  const int nbr = src->nblkrows;
  if (nbr == 0 || src->nblks == 0) return 0;
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((crop_compact<double>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const double*>(src->data), src->row_blk_size, src->col_blk_size, E->off_a.p, E->off_b.p, nbr,
                       E->crop_win, E->keep.p, E->prod_start.p, E->c_blk_p_ws.p, dst->col_i, dst->blk_p, static_cast<double*>(dst->data));
  else if (datatype == dbcsr_type_real_4)
    hipLaunchKernelGGL((crop_compact<float>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const float*>(src->data), src->row_blk_size, src->col_blk_size, E->off_a.p, E->off_b.p, nbr,
                       E->crop_win, E->keep.p, E->prod_start.p, E->c_blk_p_ws.p, dst->col_i, dst->blk_p, static_cast<float*>(dst->data));
  else
    return -10;
  return check(hipGetLastError(), "dbcsr_amd_bcsr_crop_apply", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_scale_window(void* handle, libsmm_acc_data_t datatype, dbcsr_amd_bcsr* m, double beta, int64_t row_lo, int64_t row_hi,
                                int64_t col_lo, int64_t col_hi, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !m) return -1;
  hipStream_t st = stream_of(stream);
  const int nbr = m->nblkrows;
  if (nbr == 0 || m->nblks == 0) return 0;
  const Window w = make_window(m, row_lo, row_hi, col_lo, col_hi);
  if (element_offsets(E, m->row_blk_size, nbr, E->off_a, st)) return -1;
  if (element_offsets(E, m->col_blk_size, m->nblkcols, E->off_b, st)) return -1;
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((scale_window<double>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<double*>(m->data), m->row_blk_size, m->col_blk_size, E->off_a.p, E->off_b.p, nbr, w, beta);
  else if (datatype == dbcsr_type_real_4)
    hipLaunchKernelGGL((scale_window<float>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<float*>(m->data), m->row_blk_size, m->col_blk_size, E->off_a.p, E->off_b.p, nbr, w, (float)beta);
  else
    return -10;
  return check(hipGetLastError(), "dbcsr_amd_bcsr_scale_window", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_filter_count(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, double eps, int32_t* new_row_p,
                                int64_t* new_nblks, int64_t* new_nze, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !m || !new_row_p || !new_nblks || !new_nze) return -1;
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int nbr = m->nblkrows;
  const int64_t nb = m->nblks;
  E->valid = false;  // shares workspace with the symbolic phase
  E->flt_nblks = nb;
  if (E->norms64.ensure((size_t)nb + 1) || E->keep.ensure((size_t)nb + 1) || E->blk_nze.ensure((size_t)nb + 1) ||
      E->row_nnz.ensure((size_t)nbr + 1) || E->prod_start.ensure((size_t)nb + 1) || E->c_blk_p_ws.ensure((size_t)nb + 1) ||
      E->dev_scalars.ensure(16))
    return -1;
  int64_t* dsc = reinterpret_cast<int64_t*>(E->dev_scalars.p);
  ACC_CHECK(hipMemsetAsync(dsc, 0, 16 * sizeof(int64_t), st));
  const bool have_norms = E->norms_data != nullptr && E->norms_data == m->data && E->norms_nblks == nb && datatype == dbcsr_type_real_8;
  E->norms_data = nullptr;
  if (nbr > 0 && nb > 0) {
    const int sm = row_split(nbr, nb);
    if (have_norms) {
      // left behind by the numeric kernel of the multiply that produced m
    } else if (datatype == dbcsr_type_real_8)
      hipLaunchKernelGGL((bcsr_block_norms<double>), grid_for((int64_t)nbr * sm * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                         static_cast<const double*>(m->data), m->row_blk_size, m->col_blk_size, nbr, sm, 1.0, (float*)nullptr, E->norms64.p);
    else
      hipLaunchKernelGGL((bcsr_block_norms<float>), grid_for((int64_t)nbr * sm * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                         static_cast<const float*>(m->data), m->row_blk_size, m->col_blk_size, nbr, sm, 1.0, (float*)nullptr, E->norms64.p);
    hipLaunchKernelGGL(filter_flags, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->norms64.p, nb, m->row_p, m->col_i, m->row_blk_size,
                       m->col_blk_size, nbr, eps * eps, E->keep.p, E->blk_nze.p, E->row_nnz.p);
  }
  if (exclusive_scan<int32_t>(E, E->row_nnz.p, nbr, new_row_p, dsc + 0, true, st)) return -1;
  if (exclusive_scan<int64_t>(E, E->keep.p, nb, E->prod_start.p, nullptr, false, st)) return -1;     // new index of each kept block
  if (exclusive_scan<int64_t>(E, E->blk_nze.p, nb, E->c_blk_p_ws.p, dsc + 1, false, st)) return -1;  // new data offset
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  *new_nblks = E->host_scalars[0];
  *new_nze = E->host_scalars[1];
  return check(hipGetLastError(), "dbcsr_amd_bcsr_filter_count", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_filter_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, dbcsr_amd_bcsr* dst, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !src || !dst || E->flt_nblks != src->nblks) return -1;
  hipStream_t st = stream_of(stream);
  const int nbr = src->nblkrows;
  if (nbr == 0 || src->nblks == 0) return 0;
  const int sc = row_split(nbr, src->nblks);
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((filter_compact<double>), grid_for((int64_t)nbr * sc * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const double*>(src->data), src->row_blk_size, src->col_blk_size, nbr, sc, E->keep.p, E->prod_start.p,
                       E->c_blk_p_ws.p, dst->col_i, dst->blk_p, static_cast<double*>(dst->data));
  else if (datatype == dbcsr_type_real_4)
    hipLaunchKernelGGL((filter_compact<float>), grid_for((int64_t)nbr * sc * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const float*>(src->data), src->row_blk_size, src->col_blk_size, nbr, sc, E->keep.p, E->prod_start.p,
                       E->c_blk_p_ws.p, dst->col_i, dst->blk_p, static_cast<float*>(dst->data));
  else
    return -10;
  return check(hipGetLastError(), "dbcsr_amd_bcsr_filter_apply", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_checksum(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, double* out2, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !m || !out2) return -1;
  hipStream_t st = stream_of(stream);
  const int nbr = m->nblkrows;
  out2[0] = out2[1] = 0.0;
  if (nbr == 0 || m->nblks == 0) return 0;
  if (E->row_sums.ensure((size_t)2 * nbr + 2)) return -1;
  if (element_offsets(E, m->row_blk_size, nbr, E->off_a, st)) return -1;
  if (element_offsets(E, m->col_blk_size, m->nblkcols, E->off_b, st)) return -1;
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((checksum_blocks<double>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<const double*>(m->data), m->row_blk_size, m->col_blk_size, E->off_a.p, E->off_b.p, nbr,
                       E->row_sums.p);
  else if (datatype == dbcsr_type_real_4)
    hipLaunchKernelGGL((checksum_blocks<float>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<const float*>(m->data), m->row_blk_size, m->col_blk_size, E->off_a.p, E->off_b.p, nbr,
                       E->row_sums.p);
  else
    return -10;
  hipLaunchKernelGGL(checksum_final, dim3(1), dim3(256), 0, st, E->row_sums.p, nbr, E->row_sums.p + 2 * (size_t)nbr);
  ACC_CHECK(hipMemcpyAsync(E->host_scalars + 4, E->row_sums.p + 2 * (size_t)nbr, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  memcpy(out2, E->host_scalars + 4, 2 * sizeof(double));
  return 0;
}

int dbcsr_amd_bcsr_fill_random(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, int counter, void* stream) {
  return dbcsr_amd_bcsr_fill_random_dist(handle, datatype, m, counter, nullptr, nullptr, m ? m->nblkrows : 0, stream);
}

int dbcsr_amd_bcsr_fill_random_dist(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, int counter,
                                    const int32_t* row_gid, const int32_t* col_gid, int32_t nblkrows_global, void* stream) {
  if (!handle || !m) return -1;
  hipStream_t st = stream_of(stream);
  if (m->nblks == 0) return 0;
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL(fill_random_f64, grid_for((int64_t)m->nblkrows * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<double*>(m->data), m->row_blk_size, m->col_blk_size, m->nblkrows, m->nblkcols, counter, row_gid, col_gid,
                       nblkrows_global);
  else if (datatype == dbcsr_type_real_4)
    hipLaunchKernelGGL(fill_random_f32, grid_for(m->nblks), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<float*>(m->data), m->row_blk_size, m->col_blk_size, m->nblkrows, m->nblkcols, counter, row_gid, col_gid,
                       nblkrows_global);
  else
    return -10;
  return check(hipGetLastError(), "dbcsr_amd_bcsr_fill_random", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_transpose(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, dbcsr_amd_bcsr* dst, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !src || !dst) return -1;
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int s_nbr = src->nblkrows, t_nbr = src->nblkcols;
  const int Wt = (s_nbr + 31) / 32;
  E->valid = false;  // shares workspace with the symbolic phase
  if (E->c_bm.ensure((size_t)t_nbr * Wt + 1) || E->c_pre.ensure((size_t)t_nbr * Wt + 1) || E->row_nnz.ensure((size_t)t_nbr + 1) ||
      E->blk_nze.ensure((size_t)src->nblks + 1) || E->c_blk_p_ws.ensure((size_t)src->nblks + 1))
    return -1;
  if (t_nbr == 0) return 0;
  ACC_CHECK(hipMemsetAsync(E->c_bm.p, 0, sizeof(uint32_t) * (size_t)t_nbr * Wt, st));
  if (s_nbr > 0) hipLaunchKernelGGL(transpose_mark, grid_for((int64_t)s_nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, s_nbr, Wt, E->c_bm.p);
  hipLaunchKernelGGL(row_prefix, grid_for((int64_t)t_nbr * 64), dim3(256), 0, st, E->c_bm.p, t_nbr, Wt, E->c_pre.p, E->row_nnz.p);
  if (exclusive_scan<int32_t>(E, E->row_nnz.p, t_nbr, dst->row_p, nullptr, true, st)) return -1;
  if (src->nblks > 0) {
    hipLaunchKernelGGL(transpose_sizes, grid_for((int64_t)t_nbr * Wt), dim3(256), 0, st, E->c_bm.p, E->c_pre.p, dst->row_p,
                       src->row_blk_size, src->col_blk_size, t_nbr, Wt, E->blk_nze.p);
    if (exclusive_scan<int64_t>(E, E->blk_nze.p, src->nblks, E->c_blk_p_ws.p, nullptr, false, st)) return -1;
    if (datatype == dbcsr_type_real_8)
      hipLaunchKernelGGL((transpose_fill<double>), grid_for((int64_t)s_nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                         static_cast<const double*>(src->data), src->row_blk_size, src->col_blk_size, E->c_bm.p, E->c_pre.p, dst->row_p,
                         E->c_blk_p_ws.p, s_nbr, Wt, dst->col_i, dst->blk_p, static_cast<double*>(dst->data));
    else
      hipLaunchKernelGGL((transpose_fill<float>), grid_for((int64_t)s_nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                         static_cast<const float*>(src->data), src->row_blk_size, src->col_blk_size, E->c_bm.p, E->c_pre.p, dst->row_p,
                         E->c_blk_p_ws.p, s_nbr, Wt, dst->col_i, dst->blk_p, static_cast<float*>(dst->data));
  }
  dst->nblks = src->nblks;
  return check(hipGetLastError(), "dbcsr_amd_bcsr_transpose", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_desymmetrize_count(void* handle, const dbcsr_amd_bcsr* src, int32_t* dst_row_p, int64_t* nblks, int64_t* nze, void* stream) {
  return dbcsr_amd_bcsr_twin_count(handle, src, 0, dst_row_p, nblks, nze, stream);
}

int dbcsr_amd_bcsr_desymmetrize_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, int antisymmetric, dbcsr_amd_bcsr* dst,
                                      void* stream) {
  return dbcsr_amd_bcsr_twin_apply(handle, datatype, src, 0, antisymmetric, dst, stream);
}

int dbcsr_amd_mm_set_canonical_product(void* handle, int on) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E) return -1;
  E->canonical_c = on ? 1 : 0;
  return 0;
}

int dbcsr_amd_bcsr_twin_count(void* handle, const dbcsr_amd_bcsr* src, int mode, int32_t* dst_row_p, int64_t* nblks, int64_t* nze, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !src || !dst_row_p || !nblks || !nze || src->nblkrows != src->nblkcols || mode < 0 || mode > 2) return -1;
  hipStream_t st = stream_of(stream);
  const int nbr = src->nblkrows, W = (nbr + 31) / 32;
  E->valid = false;  // shares workspace with the symbolic phase
  *nblks = *nze = 0;
  if (nbr == 0) return 0;
  if (E->c_bm.ensure((size_t)nbr * W + 1) || E->c_pre.ensure((size_t)nbr * W + 1) || E->row_nnz.ensure((size_t)nbr + 1) ||
      E->blk_nze.ensure(2 * (size_t)src->nblks + 1) || E->c_blk_p_ws.ensure(2 * (size_t)src->nblks + 1) || E->dev_scalars.ensure(16))
    return -1;
  int64_t* dsc = reinterpret_cast<int64_t*>(E->dev_scalars.p);
  ACC_CHECK(hipMemsetAsync(E->c_bm.p, 0, sizeof(uint32_t) * (size_t)nbr * W, st));
  hipLaunchKernelGGL(desym_mark, grid_for((int64_t)nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, nbr, W, mode, E->c_bm.p);
  hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->c_bm.p, nbr, W, E->c_pre.p, E->row_nnz.p);
  if (exclusive_scan<int32_t>(E, E->row_nnz.p, nbr, dst_row_p, dsc + 0, true, st)) return -1;
  // block sizes in index order (square matrix: the transposed-matrix helper with rows = columns = the same sizes)
  hipLaunchKernelGGL(transpose_sizes, grid_for((int64_t)nbr * W), dim3(256), 0, st, E->c_bm.p, E->c_pre.p, dst_row_p, src->row_blk_size,
                     src->col_blk_size, nbr, W, E->blk_nze.p);
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  *nblks = E->host_scalars[0];
  if (exclusive_scan<int64_t>(E, E->blk_nze.p, *nblks, E->c_blk_p_ws.p, dsc + 1, false, st)) return -1;
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  *nze = E->host_scalars[1];
  return check(hipGetLastError(), "dbcsr_amd_bcsr_twin_count", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_twin_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, int mode, int antisymmetric, dbcsr_amd_bcsr* dst,
                              void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !src || !dst || src->nblkrows != src->nblkcols || mode < 0 || mode > 2) return -1;
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int nbr = src->nblkrows, W = (nbr + 31) / 32;
  if (nbr == 0 || src->nblks == 0) return 0;
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((desym_fill<double>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const double*>(src->data), src->row_blk_size, E->c_bm.p, E->c_pre.p, dst->row_p, E->c_blk_p_ws.p, nbr, W,
                       antisymmetric ? -1.0 : 1.0, mode, dst->col_i, dst->blk_p, static_cast<double*>(dst->data));
  else
    hipLaunchKernelGGL((desym_fill<float>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const float*>(src->data), src->row_blk_size, E->c_bm.p, E->c_pre.p, dst->row_p, E->c_blk_p_ws.p, nbr, W,
                       antisymmetric ? -1.0f : 1.0f, mode, dst->col_i, dst->blk_p, static_cast<float*>(dst->data));
  return check(hipGetLastError(), "dbcsr_amd_bcsr_twin_apply", __FILE__, __LINE__);
}

int dbcsr_amd_mm_stats(void* handle, dbcsr_amd_mnk_stat* out, int max_entries, int* n_entries, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !n_entries || (max_entries > 0 && !out)) return -1;
  *n_entries = 0;
  if (!E->valid || !E->timed || E->c_nblks == 0) return 0;  // no numeric call yet / nothing to count
  hipStream_t st = stream_of(stream);
  if (E->stat_table.ensure(2 * (size_t)kStatSlots + 2)) return -1;
  unsigned long long* keys = E->stat_table.p;
  unsigned long long* counts = keys + kStatSlots;
  int* overflow = reinterpret_cast<int*>(counts + kStatSlots);
  ACC_CHECK(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * (2 * (size_t)kStatSlots + 2), st));
  hipLaunchKernelGGL(mnk_histogram, grid_for(E->c_nblks), dim3(256), 0, st, E->descs.p, E->c_nblks, E->entries.p, keys, counts, overflow);
  std::vector<unsigned long long> host(2 * (size_t)kStatSlots + 2);
  ACC_CHECK(hipMemcpyAsync(host.data(), keys, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  if (*reinterpret_cast<const int*>(&host[2 * (size_t)kStatSlots])) {
    fprintf(stderr, "dbcsr_amd_mm_stats: more than %d distinct (m, n, k) triples\n", kStatSlots);
    return -1;
  }
  std::vector<dbcsr_amd_mnk_stat> all;
  for (int i = 0; i < kStatSlots; ++i)
    if (host[i]) {
      dbcsr_amd_mnk_stat r;
      r.m = (int32_t)(host[i] & 0xffffu);
      r.n = (int32_t)((host[i] >> 16) & 0xffffu);
      r.k = (int32_t)((host[i] >> 32) & 0x7fffffffu);
      r.reserved = 0;
      r.nproducts = (int64_t)host[kStatSlots + i];
      r.flop = 2ll * r.m * r.n * r.k * r.nproducts;
      all.push_back(r);
    }
  std::sort(all.begin(), all.end(), [](const dbcsr_amd_mnk_stat& a, const dbcsr_amd_mnk_stat& b) {
    return a.flop != b.flop ? a.flop > b.flop : (a.m != b.m ? a.m < b.m : (a.n != b.n ? a.n < b.n : a.k < b.k));
  });
  *n_entries = (int)all.size();
  for (int i = 0; i < (int)all.size() && i < max_entries; ++i) out[i] = all[i];
  return 0;
}

const char* dbcsr_amd_mm_kernel_name(libsmm_acc_data_t datatype) {
  return datatype == dbcsr_type_real_4 ? "mm_numeric_f32" : "mm_numeric_f64";
}

const char* dbcsr_amd_mm_last_kernel(void* handle) {
  Engine* E = static_cast<Engine*>(handle);
  return E ? E->last_kernel : "";
}

int dbcsr_amd_mm_trust_plan(void* handle, int on) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E) return -1;
  E->plan_trusted = on != 0;
  return 0;
}

int dbcsr_amd_mm_plan_stats(void* handle, int64_t* reused, int64_t* built) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E) return -1;
  if (reused) *reused = E->plan_hits;
  if (built) *built = E->plan_misses;
  return 0;
}

#ifdef DBCSR_AMD_EXPERIMENTS
int dbcsr_amd_mm_tile_stats(void* handle, int* waves_gave_up, int* list_mismatches) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E) return -1;
  if (strncmp(E->last_kernel, "mm_numeric_f64_tile", 19) != 0 || !E->tile_flags.p) return 1;
  int h[4] = {0, 0, 0, 0};
  ACC_CHECK(hipDeviceSynchronize());
  ACC_CHECK(hipMemcpy(h, E->tile_flags.p, sizeof h, hipMemcpyDeviceToHost));
  if (waves_gave_up) *waves_gave_up = h[0];
  if (list_mismatches) *list_mismatches = h[1];
  if ((E->tile_knobs & 32) && E->tile_times.p) {
    unsigned long long t[8];
    ACC_CHECK(hipMemcpy(t, E->tile_times.p, sizeof t, hipMemcpyDeviceToHost));
    const double w = t[5] ? (double)t[5] : 1.0;
    fprintf(stderr, "dbcsr_amd tile kernel, mean per wave [ms]: total %.3f = window waits %.3f + operand waits %.3f + multiplies %.3f + epilogues %.3f + rest %.3f (%llu waves)\n",
            t[0] / w * 1e-5, t[1] / w * 1e-5, t[2] / w * 1e-5, t[3] / w * 1e-5, t[4] / w * 1e-5, ((double)t[0] - t[1] - t[2] - t[3] - t[4]) / w * 1e-5, t[5]);
  }
  if (getenv("DBCSR_AMD_MM_TILE_VERBOSE"))
    fprintf(stderr, "dbcsr_amd tile kernel: %d waves gave up, %lld reads of the team counters, %lld products waited for the window (of %lld)\n", h[0],
            16ll * h[2], 16ll * h[3], (long long)E->nproducts);
  return 0;
}

int dbcsr_amd_mm_band_stats(void* handle, int* waits_gave_up, int* list_mismatches) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E) return -1;
  if (strncmp(E->last_kernel, "mm_numeric_f64_band", 19) != 0 || !E->band_flags.p) return 1;
  int h[4] = {0, 0, 0, 0};
  ACC_CHECK(hipDeviceSynchronize());
  ACC_CHECK(hipMemcpy(h, E->band_flags.p, sizeof h, hipMemcpyDeviceToHost));
  if (waits_gave_up) *waits_gave_up = h[0];
  if (list_mismatches) *list_mismatches = h[1];
  if ((E->band_knobs & 1) && E->band_times.p) {
    unsigned long long t[16];
    ACC_CHECK(hipMemcpy(t, E->band_times.p, sizeof t, hipMemcpyDeviceToHost));
    const double w = t[5] ? (double)t[5] : 1.0;
    fprintf(stderr,
            "dbcsr_amd band kernel, mean per wave [ms]: total %.3f = window waits %.3f + issue and waits for A %.3f + waits for B %.3f + multiplies %.3f + "
            "epilogues %.3f + rest %.3f (%llu waves; %llu of %lld products waited for their B block, %llu fetched it themselves at the last moment; %llu "
            "fetches waited for the window, %llu reads of the team's counters, %d waves switched the throttle off; %lld list entries, shape %d, ring of %d, window %d)\n",
            t[0] / w * 1e-5, t[8] / w * 1e-5, t[1] / w * 1e-5, t[2] / w * 1e-5, t[3] / w * 1e-5, t[4] / w * 1e-5,
            ((double)t[0] - t[1] - t[2] - t[3] - t[4] - t[8]) / w * 1e-5, t[5], t[7], (long long)E->nproducts, t[6], t[9], t[10], h[3], (long long)E->band_nlist,
            E->band_shape, E->band_depth, E->band_window);
  }
  return 0;
}
#endif  // DBCSR_AMD_EXPERIMENTS

}  // extern "C"
