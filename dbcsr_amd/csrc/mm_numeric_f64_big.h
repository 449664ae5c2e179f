// mm_numeric_f64_big.h -- fp64 block products for blocks of 33 ... 80 (round 5): one WORKGROUP per C block
// Part of the device-resident multiply engine: included by mm_engine.hip after mm_numeric_f64.h.
//
// The reference tunes its kernels up to max_kernel_dim = 80 (src/core/dbcsr_config.F:185; libsmm_acc.cpp:324-339 chooses among the
// "largeDB" kernels there: operand slabs shared by the whole thread block, kernels/smm_acc_dnt_largeDB2.h:159-314).  Until round 4 such
// blocks went through mm_numeric_f64: one wave per C block walking 32 x 32 sub-tiles one after the other, every sub-tile re-reading the
// product list with fragments straight from global memory (13 TFLOP/s at 72^3).  Here:
//   * the four waves of a workgroup own the C block as a 2 x 2 arrangement of sub-blocks of TM x TN MFMA tiles (8 x 8 elements each,
//     v_mfma_f64_4x4x4_4b as everywhere in this library): up to 80 x 80, 25 accumulators per lane;
//   * a product is consumed in slabs of 16 inner indices: the slab of A (m x 16, contiguous in the column-major block) and of B (16 x n:
//     n runs of 128 bytes) are copied ONCE into LDS by all 256 threads -- bounds-checked 16-byte buffer loads whose descriptor covers
//     exactly the block, so the k tail of A arrives as zeros -- and every wave reads its fragments from there: per 4-deep k step TM + TN
//     ds_read_b64 feed TM x TN MFMAs (5 + 5 against 25: the LDS pipe is a fifth busy);
//   * slabs are double-buffered: the loads of slab i + 1 are in flight while slab i is multiplied, one barrier per slab;
//   * B's slab has a row pitch of 20 doubles (the 8 columns x 2 k of a half-wave's fragment read then fall into 16 distinct four-bank
//     groups); A's slab keeps the block's own pitch m (conflict-free for m mod 32 in [8, 24], two-way otherwise).
// Summation order per C element: products in list order (ascending k block), inside a product ascending k -- as the CPU reference.
#ifndef DBCSR_AMD_MM_NUMERIC_F64_BIG_H
#define DBCSR_AMD_MM_NUMERIC_F64_BIG_H

#include <type_traits>

namespace dbcsr_amd {

constexpr int BIG_KSL = 16;  // inner indices per slab
constexpr int BIG_PB = 20;   // pitch of B's slab in LDS, doubles
// bytes of one buffer (A part, then B part) for sub-blocks of TM x TN tiles; the A part is a whole number of 4 KiB rounds of the copy
static inline constexpr int big_a_rounds(int TM) { return (16 * TM * BIG_KSL * 8 + 4095) / 4096; }
static inline constexpr int big_b_rounds(int TN) { return (16 * TN + 31) / 32; }
static inline constexpr int big_a_bytes(int TM) { return big_a_rounds(TM) * 4096; }
// (B's part holds exactly the 16 TN columns a workgroup can own: with whole rounds of 32 columns the 5-tile shapes needed 55 KB for
// their two buffers and only two workgroups fitted a CU's 160 KB; 49 KB lets three in)
static inline constexpr int big_b_bytes(int TN) { return 16 * TN * BIG_PB * 8; }
static inline constexpr int big_lds_bytes(int TM, int TN) { return 2 * (big_a_bytes(TM) + big_b_bytes(TN)); }

template <int TM, int TN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) mm_numeric_f64_big(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                          const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                          double* __restrict__ c_out, const double* __restrict__ c_in, double alpha,
                                                          double beta, int skip_empty, const int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RA = big_a_rounds(TM), RB = big_b_rounds(TN);
  constexpr int ABYTES = big_a_bytes(TM), BUF = big_a_bytes(TM) + big_b_bytes(TN);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pos = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t cb = order[pos];
  if (cb < 0 || cb >= nblk) return;
  const Desc d = descs[cb];
  if ((skip_empty & 1) && d.prod_cnt == 0) return;
  const int m = d.m, n = d.n, cnt = d.prod_cnt;
  const Entry* e = entries + d.prod_start;
  const LaneMap L(lane);
  // this wave's sub-block: tiles [ta0, ta0 + TM) x [tc0, tc0 + TN) of the block's ceil(m / 8) x ceil(n / 8) tiles
  const int mt = (m + 7) >> 3, nt = (n + 7) >> 3;
  const int ta0 = (wid >> 1) * ((mt + 1) >> 1), tc0 = (wid & 1) * ((nt + 1) >> 1);
  const int row0 = 8 * ta0, col0 = 8 * tc0;
  double acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int c = 0; c < TN; ++c) acc[a][c] = 0.0;
  // fragment addresses inside a buffer (doubles): constants of the wave
  int fa[TM], fb[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    int row = row0 + 8 * a + L.rowl;
    row = row < m ? row : m - 1;
    fa[a] = row + m * L.kq;
  }
#pragma unroll
  for (int c = 0; c < TN; ++c) {
    int col = col0 + 8 * c + L.coll;
    col = col < n ? col : n - 1;
    fb[c] = ABYTES / 8 + col * BIG_PB + L.kq;
  }
  // copy roles: A -- thread t moves bytes [4096 r + 16 t, + 16) of the slab; B -- thread t moves k = 2 (t & 7), + 1 of column 32 r + (t >> 3)
  const int bk = 2 * (tid & 7), bc = tid >> 3;
  u32x4 ga[RA], gb[RB];
  int ks_cur = 0;  // k extent of the product whose slab sits in the registers
  int k0_cur = 0;
  auto issue = [&](uint64_t a_off, uint64_t b_off, int ks, int k0) __attribute__((always_inline)) {
    // (explicitly scalar: a descriptor the compiler believes to vary per lane turns every load into a waterfall loop, see cblock_f64_lds)
    const int abytes = __builtin_amdgcn_readfirstlane(m * ks * 8), bbytes = __builtin_amdgcn_readfirstlane(ks * n * 8);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + a_off), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + b_off), 0, bbytes, 0x00020000);
    // the whole offset travels in the VGPR operand: that one is bounds-checked whatever the generation's rule for the scalar offset is,
    // and the slab's bytes past the block's end (the k tail of A) MUST come back as zeros
    const int abase = tid * 16 + k0 * m * 8;
#pragma unroll
    for (int r = 0; r < RA; ++r) ga[r] = __builtin_amdgcn_raw_buffer_load_b128(rsa, abase + r * 4096, 0, 0);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int col = 32 * r + bc;
      // a column past the block's last one must not alias into the block: its offset is pushed past the end (the bounds check returns 0)
      const int off = col < n ? (col * ks + k0 + bk) * 8 : 0x7ffffff0;
      gb[r] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off, 0, 0);
    }
    ks_cur = ks;
    k0_cur = k0;
  };
  auto stage = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < RA; ++r) *reinterpret_cast<u32x4*>(buf + r * 4096 + tid * 16) = ga[r];
    // B: the two k of this thread that lie past the product's k extent are the next column's elements (or zeros past the block): they meet
    // A's zero padding in the MFMAs, but a NaN there must not leak into this column -- they are zeroed here
    const bool k0ok = k0_cur + bk < ks_cur, k1ok = k0_cur + bk + 1 < ks_cur;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      u32x4 v = gb[r];
      if (!k0ok) v[0] = 0u, v[1] = 0u;
      if (!k1ok) v[2] = 0u, v[3] = 0u;
      if (32 * r + bc < 16 * TN) *reinterpret_cast<u32x4*>(buf + ABYTES + ((32 * r + bc) * BIG_PB + bk) * 8) = v;
    }
  };
  // the walk over (product, slab): wave-uniform scalars
  int p = 0, k0 = 0;
  // the current product and the one after it as plain scalars (a struct handed to the lambda by reference lands in scratch memory)
  uint32_t ea = 0, eb = 0, ew = 1, na = 0, nb = 0, nw = 1;
  if (cnt > 0) ea = e[0].a_lo, eb = e[0].b_lo, ew = e[0].w;
  {
    const int i1 = cnt > 1 ? 1 : 0;
    if (cnt > 0) na = e[i1].a_lo, nb = e[i1].b_lo, nw = e[i1].w;
  }
  auto a_of = [](uint32_t lo, uint32_t w) { return (uint64_t)lo | ((uint64_t)((w >> 16) & 0xffu) << 32); };
  auto b_of = [](uint32_t lo, uint32_t w) { return (uint64_t)lo | ((uint64_t)(w >> 24) << 32); };
  if (cnt > 0) issue(a_of(ea, ew), b_of(eb, ew), (int)(ew & 0xffffu), 0);
  // The sub-blocks of the four waves are not equally large when a dimension has an odd number of tiles (72 = 9 tiles: 5 + 4; 40 = 5 tiles:
  // 3 + 2): a wave multiplies exactly the TA x TC tiles it owns, so the SIMD's matrix pipe is free for the waves of the CU's other
  // workgroups meanwhile (72^3: 81 useful tile products per step and workgroup instead of 100 issued).  The variant is chosen ONCE, outside
  // the product loop -- inside, a wave-uniform switch would put the accumulators through phi copies at every trip.
  int it = 0;
  auto run = [&](auto ta_tag, auto tc_tag) __attribute__((always_inline)) {
    constexpr int TA = decltype(ta_tag)::value, TC = decltype(tc_tag)::value;
    while (p < cnt) {
      char* buf = smem + (it & 1) * BUF;
      const int ks = ks_cur;
      const int rem = (ks - k0 + 3) >> 2;
      const int nst = rem < BIG_KSL / 4 ? rem : BIG_KSL / 4;  // k steps of this slab
      stage(buf);
      __syncthreads();
      // advance, and request the next slab while this one is multiplied
      int p2 = p, k2 = k0 + BIG_KSL;
      if (k2 >= ks) {
        p2 = p + 1;
        k2 = 0;
        ea = na, eb = nb, ew = nw;
        const int i2 = p2 + 1 < cnt ? p2 + 1 : cnt - 1;
        na = e[i2].a_lo, nb = e[i2].b_lo, nw = e[i2].w;
      }
      if (p2 < cnt) issue(a_of(ea, ew), b_of(eb, ew), (int)(ew & 0xffffu), k2);
      const double* la = reinterpret_cast<const double*>(buf);
      // the (at most four) k steps of the slab, fragments of step s + 1 requested before the MFMAs of step s
      double av[2][TA], bv[2][TC];
      auto fetch = [&](int s, int set) {
#pragma unroll
        for (int a = 0; a < TA; ++a) av[set][a] = la[fa[a] + 4 * m * s];
#pragma unroll
        for (int c = 0; c < TC; ++c) bv[set][c] = la[fb[c] + 4 * s];
      };
      fetch(0, 0);
#pragma unroll
      for (int s = 0; s < BIG_KSL / 4; ++s) {
        if (s + 1 < nst) fetch(s + 1, (s + 1) & 1);
        if (s < nst) {
#pragma unroll
          for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int c = 0; c < TC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[s & 1][a], bv[s & 1][c], acc[a][c], 0, 0, 0);
        }
      }
      p = p2;
      k0 = k2;
      ++it;
    }
    // the epilogue inside the variant too: accumulators that met again behind the switch were kept twice (214 registers for 5 x 5 tiles)
    double* C = c_out + d.c_off;
    const bool has_in = d.cin_off >= 0;
    const double* Ci = c_in + (has_in ? d.cin_off : 0);
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int c = 0; c < TC; ++c) {
        const int ta = ta0 + a, tc = tc0 + c;
        const int row = 8 * ta + L.rowd, col = 8 * tc + L.coll;
        // a tile belongs to this wave only inside its half of the tile grid (the halves overlap when a dimension has fewer tiles than 2 TM)
        const bool mine = ta < ((wid >> 1) ? mt : ((mt + 1) >> 1)) && tc < ((wid & 1) ? nt : ((nt + 1) >> 1));
        if (mine && row < m && col < n) {
          double v = alpha * acc[a][c];
          if (has_in) v += beta * Ci[row + (size_t)m * col];
          C[row + (size_t)m * col] = v;
        }
      }
  };
  {
    // tiles this wave owns in each dimension (the second half of an odd count is one tile short; a block smaller than the launch's largest
    // may own fewer still: the TM - 1 / TN - 1 variant then multiplies a few tiles nobody stores)
    const int own_r = (wid >> 1) ? mt - ((mt + 1) >> 1) : ((mt + 1) >> 1), own_c = (wid & 1) ? nt - ((nt + 1) >> 1) : ((nt + 1) >> 1);
    const int sel = (skip_empty & 4) ? 0 : (own_r >= TM ? 0 : 2) + (own_c >= TN ? 0 : 1);   // (wave-uniform; bit 2 of skip_empty: DBCSR_AMD_MM_BIG=2, every wave issues all TM x TN tile products -- measurements)
    switch (sel) {
      case 0: run(std::integral_constant<int, TM>{}, std::integral_constant<int, TN>{}); break;
      case 1: run(std::integral_constant<int, TM>{}, std::integral_constant<int, TN - 1>{}); break;
      case 2: run(std::integral_constant<int, TM - 1>{}, std::integral_constant<int, TN>{}); break;
      default: run(std::integral_constant<int, TM - 1>{}, std::integral_constant<int, TN - 1>{}); break;
    }
  }
}

}  // namespace dbcsr_amd
#endif
