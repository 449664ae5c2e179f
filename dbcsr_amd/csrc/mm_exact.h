// mm_exact.h -- exact-size fp64 block-product kernel for ONE (m, n) class of C blocks and up to three inner sizes.
//
// The reference compiles one kernel per (m, n, k) at run time and routes every parameter stack to its own kernel
// (src/acc/libsmm_acc/libsmm_acc.cpp:90-195, 281-321); stacks are homogeneous there because the host sorts products
// by the three most common sizes of each dimension (map_most_common, src/dist/dbcsr_dist_util.F:753-812).  Here a C
// block's products are summed in registers by ONE wave, so the inner size varies along its product list: the kernel
// is specialised for the block's (M, N) and carries one unrolled body per inner size K0, K1, K2 (0 = absent) -- all
// chunk counts, LDS offsets and the k loops are compile-time constants in each body (the generic kernel spends 268
// VALU + 212 SALU instructions per product next to its 54 MFMAs on BASELINE config 3,
// profiles/r02_config3_rocprofv3_pmc_summary.txt).  Products with another inner size are added straight from global
// memory after the pipelined loop.  mm_jit.hip compiles this text with hiprtc for the classes of a multiply and
// launches it once per (m, n) class on that class's segment of order[].
//
// Staging: A and B blocks of the next product are in flight in VGPRs (raw buffer loads, zero fill past the block's
// end) while the current one is multiplied from the wave's private LDS slice.  Blocks whose leading dimension is a
// multiple of 16 doubles are stored with a pitch of +2 doubles: with 128- or 256-byte column strides every column of
// a fragment read would hit the same LDS banks (SQ_LDS_BANK_CONFLICT was 48 % of the LDS cycles on config 3).
#ifndef DBCSR_AMD_MM_EXACT_H
#define DBCSR_AMD_MM_EXACT_H
#include "mm_types.h"
#include "smm_core.h"

// Tuning switches (run-time compiled kernels take them from DBCSR_AMD_JIT_DEFS, e.g. "-DDBCSR_EXACT_ALL_PIECES=1"):
//  DBCSR_EXACT_ALL_PIECES   1: every product issues the loads / LDS copies of the LARGEST inner size of the class (pieces past
//                              the block's end are bounds-checked away and copy zeros): no wave-uniform branches in the staging
//                              code; 0: piece counts follow the product's inner size (one scalar branch per piece)
//  DBCSR_EXACT_SCHED_BARRIER 1: scheduling barrier after every k step of the multiply (bounds the live fragment registers)
// Measured on BASELINE config 3 (kernel ms, all launches of a multiply; gpurun_out/s7): ALL_PIECES/SCHED_BARRIER = 0/0 9.68,
// 0/1 9.57, 1/0 9.18, 1/1 9.11 -> defaults 1 / 1.
#ifndef DBCSR_EXACT_ALL_PIECES
#define DBCSR_EXACT_ALL_PIECES 1
#endif
#ifndef DBCSR_EXACT_SCHED_BARRIER
#define DBCSR_EXACT_SCHED_BARRIER 1
#endif

namespace dbcsr_amd {

template <int LD>
struct Pitch {  // leading dimension in LDS (doubles) of a column-major block whose columns have LD elements
  static constexpr int PAD = (LD % 16 == 0) ? 2 : 0;
  static constexpr int P = LD + PAD;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
template <int V>
struct IntC {
  static constexpr int value = V;
};

// LDS bytes one wave needs for class (m, n) with inner sizes k0, k1, k2 (0 = absent): also evaluated at run time by the host
// (mm_jit.hip) to size the launch
constexpr int class_wave_lds(int m, int n, int k0, int k1, int k2) {
  int a_lds = 0, cb_max = 0, bpad = 0;
  const int ks[3] = {k0, k1, k2};
  for (int i = 0; i < 3; ++i) {
    const int k = ks[i];
    if (k <= 0) continue;
    const int k4 = 4 * ((k + 3) / 4), ap = m + ((m % 16 == 0) ? 2 : 0);
    a_lds = cmax(a_lds, ap * k4 * 8);
    cb_max = cmax(cb_max, (k * n * 8 + 1023) / 1024);
    if (k % 16 == 0) bpad = 128;
  }
  a_lds = (a_lds + 15) & ~15;
  const int c_lds = ((m * n * 8 + 1023) / 1024) * 1024;
  // B is written in whole 1 KiB pieces; with a padded pitch (columns of 16 or 32 elements) a piece spans up to 8 columns,
  // each shifted by 16 bytes more than the one before: 128 extra bytes per piece at most
  return (cmax(a_lds + cb_max * (1024 + bpad) + 16, c_lds) + 15) & ~15;
}

template <int M, int N, int K>
struct KShape {
  static constexpr int KS = (K + 3) / 4, K4 = 4 * KS;
  static constexpr int AP = Pitch<M>::P, BP = Pitch<K>::P;
  static constexpr int A_LDS = AP * K4 * 8, B_LDS = BP * N * 8;        // bytes of the staged images
  static constexpr int CA = (M * K4 * 8 + 1023) / 1024, CB = (K * N * 8 + 1023) / 1024;  // 1 KiB pieces read from global memory
};
template <int M, int N>
struct KShape<M, N, 0> {
  static constexpr int KS = 0, K4 = 0, AP = 0, BP = 0, A_LDS = 0, B_LDS = 0, CA = 0, CB = 0;
};

template <int M, int N, int K0, int K1, int K2>
struct ClassShape {
  static constexpr int MA = (M + 7) / 8, NC = (N + 7) / 8;
  static constexpr int A_LDS = (cmax(KShape<M, N, K0>::A_LDS, cmax(KShape<M, N, K1>::A_LDS, KShape<M, N, K2>::A_LDS)) + 15) & ~15;
  static constexpr int CAMAX = cmax(KShape<M, N, K0>::CA, cmax(KShape<M, N, K1>::CA, KShape<M, N, K2>::CA));
  static constexpr int CBMAX = cmax(KShape<M, N, K0>::CB, cmax(KShape<M, N, K1>::CB, KShape<M, N, K2>::CB));
  // B is written in whole 1 KiB pieces (plus the pitch padding of its columns); A's last piece may spill into B's region,
  // which is harmless: B's pieces are stored after A's (one wave, in-order LDS queue)
  static constexpr int BPAD = ((K0 % 16 == 0) || (K1 != 0 && K1 % 16 == 0) || (K2 != 0 && K2 % 16 == 0)) ? 128 : 0;
  static constexpr int B_REGION = CBMAX * (1024 + BPAD) + 16;
  static constexpr int C_LDS = ((M * N * 8 + 1023) / 1024) * 1024;
  static constexpr int WAVE_LDS = (cmax(A_LDS + B_REGION, C_LDS) + 15) & ~15;
  static_assert(WAVE_LDS == class_wave_lds(M, N, K0, K1, K2), "host and device disagree on the LDS size of a class");
};

// byte offset inside the staged image of the 16-byte granule that lane `lane` of piece `c` carries (two consecutive elements
// of a column-major block with columns of LD elements), for the padded pitch
template <int LD>
__device__ __forceinline__ int staged_offset(int c, int lane) {
  if constexpr (Pitch<LD>::PAD == 0)
    return c * 1024 + lane * 16;
  else  // 128 % LD == 0: the column index splits into a piece part and a lane part without carry
    return c * 1024 + lane * 16 + Pitch<LD>::PAD * 8 * ((c * 128) / LD + (lane * 2) / LD);
}

template <int M, int N, int K0, int K1, int K2>
__device__ __forceinline__ void cblock_f64_classes(const Desc& d, const Entry first, bool have_first, const Entry* __restrict__ entries,
                                                   const double* __restrict__ a_data,
                                                   const double* __restrict__ b_data, double* __restrict__ c_out,
                                                   const double* __restrict__ c_in, double alpha, double beta, const LaneMap& L, int lane,
                                                   char* lds, double* __restrict__ norm_out, double drop_below = 0.0) {
  typedef ClassShape<M, N, K0, K1, K2> CS;
  constexpr int MA = CS::MA, NC = CS::NC;
  constexpr int AP = Pitch<M>::P;
  char* lds_a = lds;
  char* lds_b = lds + CS::A_LDS;
  double acc[MA][NC];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = 0.0;
  const Entry* e = entries + d.prod_start;
  const int cnt = d.prod_cnt;
  const int voff = lane * 16;
  u32x4 ra[CS::CAMAX], rb[CS::CBMAX];

  // fragment addresses of A: the same for every inner size (they depend on M only)
  const double* pa[MA];
  int colc[NC];
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    int row = 8 * a + L.rowl;
    row = row < M ? row : M - 1;
    pa[a] = reinterpret_cast<const double*>(lds_a) + row + AP * L.kq;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = 8 * c + L.coll;
    colc[c] = col < N ? col : N - 1;
  }

  // product list: one vector load (lane l holds entry base + l), handed out with v_readlane
  int ebase = 0;
  uint32_t ev0 = 0, ev1 = 0, ev2 = 1;
  auto load_window = [&](int base) {
    ebase = base;
    if (cnt <= 0) return;
    const int i = base + lane < cnt ? base + lane : cnt - 1;
    ev0 = e[i].a_lo;
    ev1 = e[i].b_lo;
    ev2 = e[i].w;
  };
  load_window(0);
  auto entry_at = [&](int i) {
    if (i - ebase >= 64) load_window(i);
    const int j = __builtin_amdgcn_readfirstlane(i - ebase);
    Entry en;
    en.a_lo = (uint32_t)__builtin_amdgcn_readlane((int)ev0, j);
    en.b_lo = (uint32_t)__builtin_amdgcn_readlane((int)ev1, j);
    en.w = (uint32_t)__builtin_amdgcn_readlane((int)ev2, j);
    return en;
  };
  auto in_set = [](int ks) { return ks == K0 || (K1 != 0 && ks == K1) || (K2 != 0 && ks == K2); };
  auto next_in_set = [&](int i) {
    while (i < cnt && !in_set(entry_at(i).ks())) ++i;
    return i;
  };

  // Staging is the same code for every inner size (piece counts are wave-uniform run-time values): only the multiply
  // below is specialised per size.  (Specialising the loads and the LDS copies too makes the compiler keep several
  // staging register sets alive across the size switch: 196-274 VGPRs instead of ~130.)
  auto issue = [&](const Entry& en) {
    const int ks = en.ks();
    const int abytes = __builtin_amdgcn_readfirstlane(M * ks * 8), bbytes = __builtin_amdgcn_readfirstlane(ks * N * 8);
    const int nca = __builtin_amdgcn_readfirstlane((M * ((ks + 3) & ~3) * 8 + 1023) >> 10), ncb = (bbytes + 1023) >> 10;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + en.a_off()), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + en.b_off()), 0, bbytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < CS::CAMAX; ++c)
      if (DBCSR_EXACT_ALL_PIECES || c < nca) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CS::CBMAX; ++c)
      if (DBCSR_EXACT_ALL_PIECES || c < ncb) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
  };
  auto store = [&](int ks) {
    const int nca = __builtin_amdgcn_readfirstlane((M * ((ks + 3) & ~3) * 8 + 1023) >> 10);
    const int ncb = __builtin_amdgcn_readfirstlane((ks * N * 8 + 1023) >> 10);
#pragma unroll
    for (int c = 0; c < CS::CAMAX; ++c)
      if (DBCSR_EXACT_ALL_PIECES || c < nca) *reinterpret_cast<u32x4*>(lds_a + staged_offset<M>(c, lane)) = ra[c];
    DBCSR_AMD_LDS_ORDER();
    // B's columns have ks elements: padded pitch when ks is a multiple of 16 (then ks is 16 or 32 and divides 128)
    const int sh = __builtin_amdgcn_readfirstlane((ks & 15) == 0 ? (ks == 16 ? 4 : 5) : 31);
#pragma unroll
    for (int c = 0; c < CS::CBMAX; ++c)
      if (DBCSR_EXACT_ALL_PIECES || c < ncb) *reinterpret_cast<u32x4*>(lds_b + c * 1024 + lane * 16 + 16 * ((c * 128 + lane * 2) >> sh)) = rb[c];
  };
  auto compute_k = [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    typedef KShape<M, N, K> KSH;
    constexpr int KS = KSH::KS, BP = KSH::BP;
    const double* pb[NC];
    const double* pbt[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      pb[c] = reinterpret_cast<const double*>(lds_b) + L.kq + BP * colc[c];
      const int kt = 4 * (KS - 1) + L.kq;  // last step when K is not a multiple of 4: lanes past the end read (0, col); A's padding is zero
      pbt[c] = reinterpret_cast<const double*>(lds_b) + (kt < K ? kt : 0) + BP * colc[c];
    }
    // two-stage software pipeline over the k steps: the fragments of step s + 1 are requested before the MFMAs of step s;
    // the scheduling barrier keeps the compiler from hoisting ALL fragment reads of the product to its top (KS x (MA + NC)
    // live doubles: the multi-K kernel then needs 200-260 VGPRs and runs at one or two waves per SIMD)
    double av[2][MA], bv[2][NC];
    auto fetch = [&](int s, int buf) {
#pragma unroll
      for (int a = 0; a < MA; ++a) av[buf][a] = pa[a][s * 4 * AP];
#pragma unroll
      for (int c = 0; c < NC; ++c) bv[buf][c] = (s == KS - 1 && (K & 3)) ? pbt[c][0] : pb[c][4 * s];
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) fetch(s + 1, (s + 1) & 1);
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[s & 1][a], bv[s & 1][c], acc[a][c], 0, 0, 0);
      if (DBCSR_EXACT_SCHED_BARRIER) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // the first product: known before the list window arrives when the launch-order record carried it
  int i0;
  Entry e0;
  if (have_first && in_set(first.ks())) {
    i0 = 0;
    e0 = first;
    issue(e0);
  } else {
    i0 = next_in_set(0);
    e0 = i0 < cnt ? entry_at(i0) : Entry::make(0, 0, K0);
    if (i0 < cnt) issue(e0);
  }
  while (i0 < cnt) {
    const int kcur = e0.ks();
    const int i1 = next_in_set(i0 + 1);
    const Entry e1 = i1 < cnt ? entry_at(i1) : e0;
    store(kcur);
    if (i1 < cnt) issue(e1);
    if (kcur == K0) compute_k(IntC<K0>());
    if constexpr (K1 != 0)
      if (kcur == K1) compute_k(IntC<K1>());
    if constexpr (K2 != 0)
      if (kcur == K2) compute_k(IntC<K2>());
    i0 = i1;
    e0 = e1;
  }
  for (int p = 0; p < cnt; ++p) {
    const Entry ep = e[p];
    if (!in_set(ep.ks())) block_product_f64<MA, NC, false>(acc, a_data + ep.a_off(), b_data + ep.b_off(), M, N, ep.ks(), L);
  }

  const bool has_in = d.cin_off >= 0;
  // the final block filter is known and the block is new (mm_numeric_f64.h: cblock_f64_exact): its norm from the accumulators; a block the filter will drop is
  // neither staged nor written
  if (norm_out && drop_below > 0.0 && !has_in) {
    double s2 = 0.0;
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
        const double v = alpha * acc[a][c];
        if (row < M && col < N) s2 += v * v;
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s2 += __shfl_down(s2, off, 64);
    s2 = __shfl(s2, 0, 64);
    if (lane == 0) *norm_out = s2;
    if (s2 < drop_below) return;
    norm_out = nullptr;   // (written)
  }
  // C epilogue through LDS: the block leaves as stored, in whole 1 KiB pieces (16 B per lane), streaming hint
  constexpr int CC = (M * N * 8 + 1023) / 1024;
  double* lds_c = reinterpret_cast<double*>(lds);
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
      if (row < M && col < N) lds_c[row + M * col] = alpha * acc[a][c];
    }
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(c_out + d.c_off), 0, M * N * 8, 0x00020000);
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  double ss = 0.0;  // squared Frobenius norm of the stored block for the final filter of a filtered multiply: summed as the values leave
  // (the stores below carry the piece offset in the vector / immediate offset, not in an SGPR soffset: see cblock_f64_exact in mm_numeric_f64.h -- the store-data hazard)
  if (has_in) {
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(c_in + d.cin_off), 0, M * N * 8, 0x00020000);
    u32x4 ci[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) ci[c] = __builtin_amdgcn_raw_buffer_load_b128(rsi, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      f64x2 v = *reinterpret_cast<const f64x2*>(lds + c * 1024 + voff);
      const f64x2 w = __builtin_bit_cast(f64x2, ci[c]);
      v[0] += beta * w[0];
      v[1] += beta * w[1];
      if (norm_out) {
        const int idx = c * 128 + 2 * lane;
        if (idx < M * N) ss += v[0] * v[0];
        if (idx + 1 < M * N) ss += v[1] * v[1];
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsc, voff + c * 1024, 0, 2);
    }
  } else {
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds + c * 1024 + voff);
      if (norm_out) {
        const f64x2 x = __builtin_bit_cast(f64x2, v);
        const int idx = c * 128 + 2 * lane;
        if (idx < M * N) ss += x[0] * x[0];
        if (idx + 1 < M * N) ss += x[1] * x[1];
      }
      __builtin_amdgcn_raw_buffer_store_b128(v, rsc, voff + c * 1024, 0, 2);
    }
  }
  if (norm_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if (lane == 0) *norm_out = ss;
  }
}

// one wave per C block; order[] holds the class's segment (per-XCD streams padded with -1, as for the other kernels)
template <int M, int N, int K0, int K1, int K2>
__device__ __forceinline__ void mm_class_kernel_body(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                     const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                     double* __restrict__ c_out, const double* __restrict__ c_in, double alpha, double beta,
                                                     int skip_empty, const int* __restrict__ order, const Work* __restrict__ work,
                                                     double* __restrict__ norms, char* smem) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t pos = (int64_t)wg * (int)(blockDim.x >> 6) + wid;  // 1, 2 or 4 waves per workgroup
  Desc d;
  Entry first = Entry::make(0, 0, 0);
  bool have_first = false;
  int64_t cbi = 0;
  if (work) {  // launch-order records: descriptor and first product in one read
    const Work w = work[pos];
    if (w.prod_cnt < 0) return;
    cbi = w.cb;
    d.c_off = w.c_off, d.cin_off = w.cin_off, d.prod_start = w.prod_start, d.prod_cnt = w.prod_cnt, d.m = w.m, d.n = w.n;
    first.a_lo = w.a_lo, first.b_lo = w.b_lo, first.w = w.w;
    have_first = w.prod_cnt > 0;
  } else {
    const int64_t cb = order[pos];
    if (cb < 0 || cb >= nblk) return;
    d = descs[cb];
    cbi = cb;
  }
  const double drop_below = norms ? norms[nblk] : 0.0;   // (the final filter's threshold when the host announced it: mm_numeric_f64.h)
  if (skip_empty && d.prod_cnt == 0) return;
  const LaneMap L(lane);
  cblock_f64_classes<M, N, K0, K1, K2>(d, first, have_first, entries, a_data, b_data, c_out, c_in, alpha, beta, L, lane,
                                       smem + (size_t)wid * ClassShape<M, N, K0, K1, K2>::WAVE_LDS, norms ? norms + cbi : nullptr, drop_below);
}


// ---- G consecutive C blocks of the class per wave, the product pipeline running ACROSS block boundaries --------------------
// With few products per C block (BASELINE config 3: 3.7) a wave's life is dominated by the dependent chain
// order[] -> descriptor -> product list -> first operands (about 4 of its 9 us).  Here a wave owns G consecutive positions of
// its class segment: the G block ids and descriptors are fetched with two vector loads (lane l holds block l) and handed out
// with v_readlane, the product list of block b + 1 is requested while block b is multiplied, and the operands of the first
// product of block b + 1 are already in flight while the last product of block b is multiplied.  Same arithmetic per block
// as cblock_f64_classes (products of a block in list order, then the products of other inner sizes), so results are identical.
template <int M, int N, int K0, int K1, int K2, int G>
__device__ __forceinline__ void mm_class_stream_body(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                     const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                     double* __restrict__ c_out, const double* __restrict__ c_in, double alpha, double beta,
                                                     int skip_empty, const int* __restrict__ order, char* smem) {
  static_assert(G >= 1 && G <= 64, "one lane per block of the group");
  typedef ClassShape<M, N, K0, K1, K2> CS;
  constexpr int MA = CS::MA, NC = CS::NC;
  constexpr int AP = Pitch<M>::P;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t pos0 = ((int64_t)wg * 4 + wid) * G;
  char* lds = smem + (size_t)wid * CS::WAVE_LDS;
  char* lds_a = lds;
  char* lds_b = lds + CS::A_LDS;
  const LaneMap L(lane);
  const int voff = lane * 16;

  // block ids and descriptors of the group: lane l < G holds block l
  const int my_cb = lane < G ? order[pos0 + lane] : -1;
  const bool my_ok = my_cb >= 0 && my_cb < nblk;
  const Desc my_d = descs[my_ok ? my_cb : 0];
  const int my_cnt = my_ok ? my_d.prod_cnt : -1;  // -1: no block at this position
  auto lane_i32 = [&](int v, int l) { return __builtin_amdgcn_readlane(v, l); };
  auto lane_i64 = [&](int64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
  };

  const double* pa[MA];
  int colc[NC];
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    int row = 8 * a + L.rowl;
    row = row < M ? row : M - 1;
    pa[a] = reinterpret_cast<const double*>(lds_a) + row + AP * L.kq;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = 8 * c + L.coll;
    colc[c] = col < N ? col : N - 1;
  }
  u32x4 ra[CS::CAMAX], rb[CS::CBMAX];
  double acc[MA][NC];

  // product-list windows (first 64 entries of a block's list), two of them: the block being multiplied and the next one
  uint32_t w0[2] = {0, 0}, w1[2] = {0, 0}, w2[2] = {1, 1};
  auto load_window = [&](int buf, int b) {  // b wave-uniform
    const int cnt = lane_i32(my_cnt, b);
    if (cnt <= 0) return;
    const Entry* e = entries + lane_i64(my_d.prod_start, b);
    const int i = lane < cnt ? lane : cnt - 1;
    const uint32_t x0 = e[i].a_lo, x1 = e[i].b_lo, x2 = e[i].w;
    if (buf) {  // (no run-time index into the register arrays: that would put them into scratch)
      w0[1] = x0, w1[1] = x1, w2[1] = x2;
    } else {
      w0[0] = x0, w1[0] = x1, w2[0] = x2;
    }
  };
  auto entry_at = [&](int buf, int b, int i) {  // entry i of block b; beyond the window: straight from memory
    Entry en;
    if (i < 64) {
      en.a_lo = (uint32_t)__builtin_amdgcn_readlane((int)(buf ? w0[1] : w0[0]), i);
      en.b_lo = (uint32_t)__builtin_amdgcn_readlane((int)(buf ? w1[1] : w1[0]), i);
      en.w = (uint32_t)__builtin_amdgcn_readlane((int)(buf ? w2[1] : w2[0]), i);
    } else {
      en = entries[lane_i64(my_d.prod_start, b) + i];
    }
    return en;
  };
  auto in_set = [](int ks) { return ks == K0 || (K1 != 0 && ks == K1) || (K2 != 0 && ks == K2); };

  auto issue = [&](const Entry& en) {
    const int ks = en.ks();
    const int abytes = __builtin_amdgcn_readfirstlane(M * ks * 8), bbytes = __builtin_amdgcn_readfirstlane(ks * N * 8);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + en.a_off()), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + en.b_off()), 0, bbytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < CS::CAMAX; ++c) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CS::CBMAX; ++c) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
  };
  auto store = [&](int ks) {
#pragma unroll
    for (int c = 0; c < CS::CAMAX; ++c) *reinterpret_cast<u32x4*>(lds_a + staged_offset<M>(c, lane)) = ra[c];
    DBCSR_AMD_LDS_ORDER();
    const int sh = __builtin_amdgcn_readfirstlane((ks & 15) == 0 ? (ks == 16 ? 4 : 5) : 31);
#pragma unroll
    for (int c = 0; c < CS::CBMAX; ++c) *reinterpret_cast<u32x4*>(lds_b + c * 1024 + lane * 16 + 16 * ((c * 128 + lane * 2) >> sh)) = rb[c];
  };
  auto compute_k = [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    typedef KShape<M, N, K> KSH;
    constexpr int KS = KSH::KS, BP = KSH::BP;
    const double* pb[NC];
    const double* pbt[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      pb[c] = reinterpret_cast<const double*>(lds_b) + L.kq + BP * colc[c];
      const int kt = 4 * (KS - 1) + L.kq;
      pbt[c] = reinterpret_cast<const double*>(lds_b) + (kt < K ? kt : 0) + BP * colc[c];
    }
    double av[2][MA], bv[2][NC];
    auto fetch = [&](int s, int buf) {
#pragma unroll
      for (int a = 0; a < MA; ++a) av[buf][a] = pa[a][s * 4 * AP];
#pragma unroll
      for (int c = 0; c < NC; ++c) bv[buf][c] = (s == KS - 1 && (K & 3)) ? pbt[c][0] : pb[c][4 * s];
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) fetch(s + 1, (s + 1) & 1);
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[s & 1][a], bv[s & 1][c], acc[a][c], 0, 0, 0);
      if (DBCSR_EXACT_SCHED_BARRIER) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // cursor over the in-set products of the group: (block, index); "active" blocks only (a block skipped by skip_empty or an
  // empty position has no products and no epilogue)
  auto block_cnt = [&](int b) { return b < G ? lane_i32(my_cnt, b) : -1; };
  auto first_in_set = [&](int buf, int b, int from) {
    const int cnt = block_cnt(b);
    int i = from;
    while (i < cnt && !in_set(entry_at(buf, b, i).ks())) ++i;
    return i < cnt ? i : -1;
  };

  load_window(0, 0);
  // next product to issue: search forward from block 0
  int nb = 0, ni = first_in_set(0, 0, 0);  // (nb, ni) = the product whose operands are in flight; ni = -1: block nb has none (left)
  bool in_flight = false;
  Entry ne = Entry::make(0, 0, K0);
  if (ni >= 0) {
    ne = entry_at(0, 0, ni);
    issue(ne);
    in_flight = true;
  }
  for (int b = 0; b < G; ++b) {
    const int buf = b & 1;
    const int cnt = block_cnt(b);
    if (b + 1 < G) load_window(buf ^ 1, b + 1);  // the next block's list travels while this block is multiplied
    if (cnt < 0 || (skip_empty && cnt == 0)) {
      // nothing to do for this position; if no product is in flight, look into the next block for one
      if (!in_flight && b + 1 < G) {
        ni = first_in_set(buf ^ 1, b + 1, 0);
        nb = b + 1;
        if (ni >= 0) {
          ne = entry_at(buf ^ 1, b + 1, ni);
          issue(ne);
          in_flight = true;
        }
      }
      continue;
    }
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[a][c] = 0.0;
    // the product in flight, if any, belongs to this block (the search never runs past a block that still has an epilogue to do)
    while (in_flight && nb == b) {
      const int kcur = ne.ks();
      store(kcur);
      // next in-set product: in this block, else the first one of the next block
      int i1 = first_in_set(buf, b, ni + 1);
      in_flight = false;
      if (i1 >= 0) {
        ni = i1;
        ne = entry_at(buf, b, i1);
        issue(ne);
        in_flight = true;
      } else if (b + 1 < G) {
        i1 = first_in_set(buf ^ 1, b + 1, 0);
        nb = b + 1;
        ni = i1;
        if (i1 >= 0) {
          ne = entry_at(buf ^ 1, b + 1, i1);
          issue(ne);
          in_flight = true;
        }
      }
      if (kcur == K0) compute_k(IntC<K0>());
      if constexpr (K1 != 0)
        if (kcur == K1) compute_k(IntC<K1>());
      if constexpr (K2 != 0)
        if (kcur == K2) compute_k(IntC<K2>());
    }
    if (!in_flight && nb <= b && b + 1 < G) {  // this block had no in-set product at all: start the next block's first one now
      ni = first_in_set(buf ^ 1, b + 1, 0);
      nb = b + 1;
      if (ni >= 0) {
        ne = entry_at(buf ^ 1, b + 1, ni);
        issue(ne);
        in_flight = true;
      }
    }
    const int64_t d_c_off = lane_i64(my_d.c_off, b), d_cin_off = lane_i64(my_d.cin_off, b), d_ps = lane_i64(my_d.prod_start, b);
    for (int p = 0; p < cnt; ++p) {  // products of another inner size: straight from global memory
      const Entry ep = entries[d_ps + p];
      if (!in_set(ep.ks())) block_product_f64<MA, NC, false>(acc, a_data + ep.a_off(), b_data + ep.b_off(), M, N, ep.ks(), L);
    }
    // C epilogue through LDS (the operands of the next product are still in registers, not in LDS)
    constexpr int CC = (M * N * 8 + 1023) / 1024;
    double* lds_c = reinterpret_cast<double*>(lds);
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
        if (row < M && col < N) lds_c[row + M * col] = alpha * acc[a][c];
      }
    const bool has_in = d_cin_off >= 0;
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(c_out + d_c_off), 0, M * N * 8, 0x00020000);
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    if (has_in) {
      const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(c_in + d_cin_off), 0, M * N * 8, 0x00020000);
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        f64x2 v = *reinterpret_cast<const f64x2*>(lds + c * 1024 + voff);
        const f64x2 w = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rsi, voff, c * 1024, 0));
        v[0] += beta * w[0];
        v[1] += beta * w[1];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsc, voff + c * 1024, 0, 2);
      }
    } else {
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(lds + c * 1024 + voff);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsc, voff + c * 1024, 0, 2);
      }
    }
  }
}

}  // namespace dbcsr_amd

#ifdef DBCSR_AMD_JIT_M  // translation unit handed to hiprtc (mm_jit.hip): one kernel, its shape comes from the macros
#ifndef DBCSR_AMD_JIT_MINW
#define DBCSR_AMD_JIT_MINW 1
#endif
extern "C" __global__ void __launch_bounds__(256, DBCSR_AMD_JIT_MINW)  // second argument: waves per SIMD the register allocation must allow
    mm_numeric_f64_class(const dbcsr_amd::Desc* __restrict__ descs, long nblk, const dbcsr_amd::Entry* __restrict__ entries,
                         const double* __restrict__ a_data, const double* __restrict__ b_data, double* __restrict__ c_out,
                         const double* __restrict__ c_in, double alpha, double beta, int skip_empty, const int* __restrict__ order,
                         const dbcsr_amd::Work* __restrict__ work, double* __restrict__ norms) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#if defined(DBCSR_AMD_JIT_G) && DBCSR_AMD_JIT_G > 1
  dbcsr_amd::mm_class_stream_body<DBCSR_AMD_JIT_M, DBCSR_AMD_JIT_N, DBCSR_AMD_JIT_K0, DBCSR_AMD_JIT_K1, DBCSR_AMD_JIT_K2, DBCSR_AMD_JIT_G>(
      descs, nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order, smem);
#else
  dbcsr_amd::mm_class_kernel_body<DBCSR_AMD_JIT_M, DBCSR_AMD_JIT_N, DBCSR_AMD_JIT_K0, DBCSR_AMD_JIT_K1, DBCSR_AMD_JIT_K2>(
      descs, nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order, work, norms, smem);
#endif
}
#endif
#endif
