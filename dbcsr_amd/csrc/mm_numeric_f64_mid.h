// mm_numeric_f64_mid.h -- fp64 block products for blocks of 33 ... 40 (round 6): ONE WAVE per C block, operands in slabs of 8 inner indices
// Part of the device-resident multiply engine: included by mm_engine.hip after mm_numeric_f64_big.h.
//
// Round 5 gave the blocks of 33 ... 80 a workgroup per C block (mm_numeric_f64_big.h: 2 x 2 waves, slabs shared through LDS): 0.59-0.65 of the fp64 peak
// from 64 on, but 0.29 at 33^3 and 0.47 at 40^3.  Measured in round 6 (profiles/r06_big_blocks_sub4_experiment.txt): at these sizes that kernel is bound
// by the CU's LDS -- a wave's sub-block of 3 x 3 or 2 x 2 tiles reads 0.67-1.0 operand registers per MFMA, and a finer partition only adds reads.  So the
// small end of the range goes the other way: one wave owns the WHOLE C block (up to 40 x 40: 25 accumulators, 10 operand reads per 25 MFMAs), nothing is
// shared, no barrier.  The reference tunes a kernel per triplet up to 80 (src/core/dbcsr_config.F:185, libsmm_acc.cpp:198-253, 324-339).
//   * a product is consumed in slabs of 8 inner indices: the slab of A (m x 8, contiguous in the column-major block) and of B (8 x n: n runs of 64
//     bytes) arrive with bounds-checked 16-byte buffer loads whose descriptor covers exactly the block -- the k tail of A comes back as zeros --, are
//     copied into the wave's LDS slice and read from there; the next slab is in flight in registers meanwhile (one LDS buffer: a wave's LDS operations
//     complete in order, the copy of slab i + 1 cannot overtake the fragment reads of slab i);
//   * the block is covered in units of 4 x 4 (BigSub): pairs of blocks as 2 x 2 arrangements of v_mfma_f64_4x4x4_4b, an odd last block row / column as
//     1 x 4 / 4 x 1 arrangements -- 33 ... 36 take 21 instructions per k step instead of the 25 of a block padded to 40, 37 ... 40 take 25;
//   * B's slab has a column pitch of 12 doubles (the 8 columns x 2 k of a half-wave's fragment read fall into 16 distinct two-bank groups).
// Summation order per C element: products in list order, inside a product ascending k -- as the CPU reference and the other kernels.
#ifndef DBCSR_AMD_MM_NUMERIC_F64_MID_H
#define DBCSR_AMD_MM_NUMERIC_F64_MID_H

#include <type_traits>

namespace dbcsr_amd {

// ---- a wave's sub-block in units of 4 x 4 -----------------------------------------------------------------------------------------------------------
// v_mfma_f64_4x4x4_4b multiplies four INDEPENDENT 4 x 4 x 4 products; which four is only a matter of the addresses its lanes read their operands
// from.  A sub-block of RB x CB blocks of 4 x 4 is covered by
//   * floor(RB / 2) x floor(CB / 2) instructions in the 2 x 2 arrangement (an 8 x 8 tile: one A register per row pair, one B register per column pair),
//   * RB odd: the last block row as 1 x 4 arrangements -- the same 4 rows against 4 consecutive column blocks per instruction, corner included,
//   * CB odd: the last block column as 4 x 1 arrangements over the remaining rows.
template <int RB, int CB>
struct BigSub {
  static constexpr int PA = RB / 2, PC = CB / 2, ODDR = RB & 1, ODDC = CB & 1;
  static constexpr int RBE = RB - ODDR;                                          // block rows the right edge covers (the corner belongs to the bottom edge)
  static constexpr int NEB = ODDR ? (CB + 3) / 4 : 0, NER = ODDC ? (RBE + 3) / 4 : 0;  // edge instructions per k step
  static constexpr int NA = PA + ODDR + NER, NB = PC + ODDC + NEB;               // operand registers per k step
  static constexpr int NMFMA = PA * PC + NEB + NER;
  // operand sets in flight per k step: two (the fragments of step s + 1 are requested before the MFMAs of step s) while the registers allow it; the
  // largest sub-blocks (10 x 10 blocks: 25 accumulators + 20 operands, 9 x 9: 21 + 15) fetch a step's operands right before its MFMAs and leave the
  // LDS latency to the SIMD's other waves -- with two sets the kernel would not fit the 168 registers of three waves per SIMD
  // (measured, session r06_11: 9 x 9 with two sets 13.1 ms against 12.3 at 33^3; with one set squeezed into the 128 registers of four waves per SIMD 13.6)
  static constexpr int SETS = (2 * NMFMA + 4 * (NA + NB) <= 100) ? 2 : 1;
  double acc[PA > 0 ? PA : 1][PC > 0 ? PC : 1], accb[NEB > 0 ? NEB : 1], accr[NER > 0 ? NER : 1];
  int fa[NA > 0 ? NA : 1], fb[NB > 0 ? NB : 1];   // offsets (doubles) of this lane's operand elements inside a slab buffer, k step 0

  // (row0, col0): the sub-block's first row / column; the A slab has the block's own pitch m; B's element (k, col) sits at bbase + col * bcs + k * bks
  __device__ __forceinline__ void init(int row0, int col0, int m, int n, int lane, int bbase, int bcs, int bks) {
    const int kq = lane >> 4, blk = (lane >> 2) & 3, x = lane & 3;
    const int p = blk >> 1, q = blk & 1;
    auto arow = [&](int r) { return (r < m ? r : m - 1) + m * kq; };
    auto bcol = [&](int c) { return bbase + (c < n ? c : n - 1) * bcs + kq * bks; };
#pragma unroll
    for (int a = 0; a < PA; ++a) fa[a] = arow(row0 + 8 * a + 4 * p + x);
    if constexpr (ODDR) fa[PA] = arow(row0 + 4 * (RB - 1) + x);
#pragma unroll
    for (int j = 0; j < NER; ++j) fa[PA + ODDR + j] = arow(row0 + 16 * j + 4 * blk + x);
#pragma unroll
    for (int c = 0; c < PC; ++c) fb[c] = bcol(col0 + 8 * c + 4 * q + x);
    if constexpr (ODDC) fb[PC] = bcol(col0 + 4 * (CB - 1) + x);
#pragma unroll
    for (int j = 0; j < NEB; ++j) fb[PC + ODDC + j] = bcol(col0 + 16 * j + 4 * blk + x);
#pragma unroll
    for (int a = 0; a < PA; ++a)
#pragma unroll
      for (int c = 0; c < PC; ++c) acc[a][c] = 0.0;
#pragma unroll
    for (int j = 0; j < NEB; ++j) accb[j] = 0.0;
#pragma unroll
    for (int j = 0; j < NER; ++j) accr[j] = 0.0;
  }
  __device__ __forceinline__ void fetch(const double* la, int aoff, int boff, double (&av)[NA > 0 ? NA : 1], double (&bv)[NB > 0 ? NB : 1]) const {
#pragma unroll
    for (int a = 0; a < NA; ++a) av[a] = la[fa[a] + aoff];
#pragma unroll
    for (int c = 0; c < NB; ++c) bv[c] = la[fb[c] + boff];
  }
  __device__ __forceinline__ void mma(const double (&av)[NA > 0 ? NA : 1], const double (&bv)[NB > 0 ? NB : 1]) {
#pragma unroll
    for (int a = 0; a < PA; ++a)
#pragma unroll
      for (int c = 0; c < PC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
    if constexpr (ODDR) {
#pragma unroll
      for (int j = 0; j < NEB; ++j) accb[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[PA], bv[PC + ODDC + j], accb[j], 0, 0, 0);
    }
    if constexpr (ODDC) {
#pragma unroll
      for (int j = 0; j < NER; ++j) accr[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[PA + ODDR + j], bv[PC], accr[j], 0, 0, 0);
    }
  }
  // f(row, col, sum) for every element of the sub-block this lane holds, rows / columns relative to (row0, col0) in [0, 4 own_r) x [0, 4 own_c) only
  // (a wave whose block is smaller than the variant's sub-block multiplied a few blocks nobody stores); zeroes the sums
  template <class F>
  __device__ __forceinline__ void drain(int lane, int own_r, int own_c, F f) {
    visit<true>(lane, own_r, own_c, f);
  }
  // ZERO = false: the sums stay (a look at them before they are drained: the block's norm for an announced filter)
  template <bool ZERO, class F>
  __device__ __forceinline__ void visit(int lane, int own_r, int own_c, F f) {
    const int kq = lane >> 4, blk = (lane >> 2) & 3, x = lane & 3;
    const int p = blk >> 1, q = blk & 1;
    const int rlim = 4 * own_r, clim = 4 * own_c;
#pragma unroll
    for (int a = 0; a < PA; ++a)
#pragma unroll
      for (int c = 0; c < PC; ++c) {
        const int r = 8 * a + 4 * p + kq, cc = 8 * c + 4 * q + x;
        if (r < rlim && cc < clim) f(r, cc, acc[a][c]);
        if (ZERO) acc[a][c] = 0.0;
      }
#pragma unroll
    for (int j = 0; j < NEB; ++j) {
      const int r = 4 * (RB - 1) + kq, cc = 16 * j + 4 * blk + x;
      if (4 * j + blk < CB && r < rlim && cc < clim) f(r, cc, accb[j]);
      if (ZERO) accb[j] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < NER; ++j) {
      const int r = 16 * j + 4 * blk + kq, cc = 4 * (CB - 1) + x;
      if (4 * j + blk < RBE && r < rlim && cc < clim) f(r, cc, accr[j]);
      if (ZERO) accr[j] = 0.0;
    }
  }
};


// KSL inner indices per slab (8 or 16); a column of B's slab has a pitch of KSL + 4 doubles in LDS
static inline constexpr int mid_a_bytes(int TM, int KSL) { return 8 * TM * KSL * 8; }           // a block has at most 8 TM rows
static inline constexpr int mid_b_bytes(int TN, int KSL) { return 8 * TN * (KSL + 4) * 8; }     // ... and 8 TN columns
// (A's copy moves whole KiB: its last round may run into B's part, which is stored after it)
static inline constexpr int mid_lds_bytes(int TM, int TN, int KSL) {
  const int a = mid_a_bytes(TM, KSL), ar = ((a + 1023) / 1024) * 1024, ab = a + mid_b_bytes(TN, KSL);
  return ar > ab ? ar : ab;
}

// RBX x CBX: the C blocks this launch multiplies, in units of 4 x 4 (6 ... 12 per dimension), exactly.  A multiply whose dominant block is not the
// largest shape (10 x 10 up to 40, 12 x 12 up to 48) takes two launches over the same order[]: <RBX, CBX> for the dominant blocks, then the largest shape -- which covers any block of the multiply --
// with bit 4 of `flags` for all the others (the matrix's tail row / column, other sizes of a mix); a wave that finds a block of the other launch leaves
// after reading its descriptor.  (Two variants behind a wave-uniform branch in ONE kernel keep both register sets live side by side under LLVM's CFG
// structurizer -- 169 registers where either needs 144 or fewer -- and nine variants spill.)
// flags: bit 0: C blocks without products stay as they are (in-place accumulation); bit 4: the launch for "all the others" -- skip the blocks of
// (bits 8-11) x (bits 12-15) units; bit 5: the only launch -- take every block (the dominant size IS the largest shape); neither: blocks of RBX x CBX only
template <int RBX, int CBX, int MID_KSL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) mm_numeric_f64_mid(const Desc* __restrict__ descs, int64_t nblk,
                                                          const Entry* __restrict__ entries, const double* __restrict__ a_data,
                                                          const double* __restrict__ b_data, double* __restrict__ c_out,
                                                          const double* __restrict__ c_in, double alpha, double beta, int flags,
                                                          const int* __restrict__ order, const Work* __restrict__ work, double* __restrict__ norms) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // rounds of the slab copies: A -- the wave moves 1 KiB per round; B -- a lane moves two k of a column: 128 / KSL columns per round
  constexpr int TM = (RBX + 1) / 2, TN = (CBX + 1) / 2;   // pairs of 4 x 4 blocks per dimension the LDS slice is sized for
  constexpr int MID_PB = MID_KSL + 4, KL = MID_KSL / 2, CPR = 64 / KL;
  constexpr int ABYTES = mid_a_bytes(TM, MID_KSL);
  constexpr int RA = (ABYTES + 1023) / 1024, RB_ = (8 * TN + CPR - 1) / CPR;
  const int lane = threadIdx.x;
  const int pos = xcd_remap(blockIdx.x, gridDim.x);
  Desc d;
  uint32_t fa_lo = 0, fb_lo = 0, fw = 1;   // the block's first product, when the launch-order record carried it
  bool have_first = false;
  int64_t cb_index = 0;   // the C block (its squared norm goes to norms[cb_index] when a filtered multiply asks for it)
  if (work) {   // launch-order records (build_work): descriptor and first product in one read -- no order[] -> descs[] -> entries[] chain
    const Work w = work[pos];
    if (w.prod_cnt < 0) return;  // padding position
    d.c_off = w.c_off, d.cin_off = w.cin_off, d.prod_start = w.prod_start, d.prod_cnt = w.prod_cnt, d.m = w.m, d.n = w.n;
    fa_lo = w.a_lo, fb_lo = w.b_lo, fw = w.w;
    have_first = w.prod_cnt > 0;
    cb_index = w.cb;
  } else {
    const int64_t cb = order[pos];
    if (cb < 0 || cb >= nblk) return;
    d = descs[cb];
    cb_index = cb;
  }
  if ((flags & 1) && d.prod_cnt == 0) return;
  const int m = __builtin_amdgcn_readfirstlane((int)d.m), n = __builtin_amdgcn_readfirstlane((int)d.n), cnt = __builtin_amdgcn_readfirstlane(d.prod_cnt);
  const int own_r = (m + 3) >> 2, own_c = (n + 3) >> 2;   // the block in units of 4 x 4
  if (flags & 16) {
    if (own_r == ((flags >> 8) & 15) && own_c == ((flags >> 12) & 15)) return;   // the exact launch multiplied it
  } else if (!(flags & 32)) {
    if (own_r != RBX || own_c != CBX) return;                                      // the other launch multiplies it
  }
  const Entry* e = entries + d.prod_start;
  // copy roles: A -- lane l moves bytes [1024 r + 16 l, + 16) of the slab; B -- lane l moves k = 2 (l mod KL), + 1 of column CPR r + l / KL
  const int bk = 2 * (lane & (KL - 1)), bc = lane / KL;
  u32x4 ga[RA], gb[RB_];
  int ks_cur = 0, k0_cur = 0;
  auto issue = [&](uint64_t a_off, uint64_t b_off, int ks, int k0) __attribute__((always_inline)) {
    const int abytes = __builtin_amdgcn_readfirstlane(m * ks * 8), bbytes = __builtin_amdgcn_readfirstlane(ks * n * 8);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + a_off), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + b_off), 0, bbytes, 0x00020000);
    // the whole offset travels in the bounds-checked VGPR operand: the slab's bytes past the block's end (the k tail of A) MUST come back as zeros
    const int abase = lane * 16 + k0 * m * 8;
#pragma unroll
    for (int r = 0; r < RA; ++r) ga[r] = __builtin_amdgcn_raw_buffer_load_b128(rsa, abase + r * 1024, 0, 0);
#pragma unroll
    for (int r = 0; r < RB_; ++r) {
      const int col = CPR * r + bc;
      // a column past the block's last one must not alias into the block: its offset is pushed past the end (the bounds check returns 0)
      const int off = col < n ? (col * ks + k0 + bk) * 8 : 0x7ffffff0;
      gb[r] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off, 0, 0);
    }
    ks_cur = ks;
    k0_cur = k0;
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < RA; ++r) *reinterpret_cast<u32x4*>(smem + r * 1024 + lane * 16) = ga[r];
    DBCSR_AMD_LDS_ORDER();
    // B: the k of this lane that lie past the product's k extent are the next column's elements (or zeros past the block): they meet A's zero
    // padding in the MFMAs, but a NaN there must not leak into this column -- they are zeroed here
    const bool k0ok = k0_cur + bk < ks_cur, k1ok = k0_cur + bk + 1 < ks_cur;
#pragma unroll
    for (int r = 0; r < RB_; ++r) {
      u32x4 v = gb[r];
      if (!k0ok) v[0] = 0u, v[1] = 0u;
      if (!k1ok) v[2] = 0u, v[3] = 0u;
      if (CPR * r + bc < 8 * TN) *reinterpret_cast<u32x4*>(smem + ABYTES + ((CPR * r + bc) * MID_PB + bk) * 8) = v;
    }
  };
  int p = 0, k0 = 0;
  uint32_t ea = 0, eb = 0, ew = 1, na = 0, nb = 0, nw = 1;   // the current product and the one after it (plain scalars: a struct handed to the lambda lands in scratch)
  if (have_first)
    ea = fa_lo, eb = fb_lo, ew = fw;
  else if (cnt > 0)
    ea = e[0].a_lo, eb = e[0].b_lo, ew = e[0].w;
  ea = (uint32_t)__builtin_amdgcn_readfirstlane((int)ea), eb = (uint32_t)__builtin_amdgcn_readfirstlane((int)eb), ew = (uint32_t)__builtin_amdgcn_readfirstlane((int)ew);
  auto a_of = [](uint32_t lo, uint32_t w) { return (uint64_t)lo | ((uint64_t)((w >> 16) & 0xffu) << 32); };
  auto b_of = [](uint32_t lo, uint32_t w) { return (uint64_t)lo | ((uint64_t)(w >> 24) << 32); };
  if (cnt > 0) issue(a_of(ea, ew), b_of(eb, ew), (int)(ew & 0xffffu), 0);
  {   // (the second product's record: requested after the first operands are on their way)
    const int i1 = cnt > 1 ? 1 : 0;
    if (cnt > 0) na = e[i1].a_lo, nb = e[i1].b_lo, nw = e[i1].w;
  }
  // (a block smaller than RBX x CBX units -- only under <10, 10> -- multiplies a few blocks of 4 x 4 nobody stores)
  {
    typedef BigSub<RBX, CBX> Sub;
    Sub S;
    S.init(0, 0, m, n, lane, ABYTES / 8, MID_PB, 1);
    const double* la = reinterpret_cast<const double*>(smem);
    while (p < cnt) {
      const int ks = ks_cur;
      const int rem = (ks - k0 + 3) >> 2;
      const int nst = rem < MID_KSL / 4 ? rem : MID_KSL / 4;  // k steps of this slab
      stage();
      // advance, and request the next slab while this one is multiplied
      int p2 = p, k2 = k0 + MID_KSL;
      if (k2 >= ks) {
        p2 = p + 1;
        k2 = 0;
        ea = na, eb = nb, ew = nw;
        const int i2 = p2 + 1 < cnt ? p2 + 1 : cnt - 1;
        na = e[i2].a_lo, nb = e[i2].b_lo, nw = e[i2].w;
      }
      if (p2 < cnt) issue(a_of(ea, ew), b_of(eb, ew), (int)(ew & 0xffffu), k2);
      double av[Sub::SETS][Sub::NA > 0 ? Sub::NA : 1], bv[Sub::SETS][Sub::NB > 0 ? Sub::NB : 1];
      if constexpr (Sub::SETS == 2) S.fetch(la, 0, 0, av[0], bv[0]);
#pragma unroll
      for (int s = 0; s < MID_KSL / 4; ++s) {
        if constexpr (Sub::SETS == 2) {
          if (s + 1 < nst) S.fetch(la, 4 * m * (s + 1), 4 * (s + 1), av[(s + 1) & 1], bv[(s + 1) & 1]);
          if (s < nst) S.mma(av[s & 1], bv[s & 1]);
        } else if (s < nst) {
          S.fetch(la, 4 * m * s, 4 * s, av[0], bv[0]);
          S.mma(av[0], bv[0]);
          __builtin_amdgcn_sched_barrier(0);   // (keeps the next step's fragment reads -- a second operand set -- behind this step's MFMAs)
        }
      }
      p = p2;
      k0 = k2;
    }
    double ss = 0.0;
    // the final block filter announced (norms[nblk] = its eps^2; mm_numeric_f64.h: cblock_f64_exact): a new block's norm from the accumulators, and a block the
    // filter is going to drop is neither staged nor written
    if (norms && d.cin_off < 0) {
      const double drop_below = norms[nblk];
      if (drop_below > 0.0) {
        double s2 = 0.0;
        S.template visit<false>(lane, own_r, own_c, [&](int row, int col, double sum) {
          const double v = alpha * sum;
          if (row < m && col < n) s2 += v * v;
        });
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s2 += __shfl_down(s2, off, 64);
        s2 = __shfl(s2, 0, 64);
        if (lane == 0) norms[cb_index] = s2;
        if (s2 < drop_below) return;
        norms = nullptr;   // (written)
      }
    }
    if constexpr (RBX <= 8 && CBX <= 8) {
      // C epilogue through LDS (as the exact-size kernels, mm_numeric_f64.h): the block is laid out as stored (column-major, contiguous) in the wave's
      // slice -- up to 32 x 32: it fits the slabs' 9 KB -- and leaves in whole 1 KiB pieces, 16 bytes per lane, with the streaming hint.  These shapes
      // are the classes of a mixed-size multiply with FEW products per C block (config 3: 3.6), where C's traffic counts: 7.25 against 7.6 ms there.
      double* lds_c = reinterpret_cast<double*>(smem);
      S.drain(lane, own_r, own_c, [&](int row, int col, double sum) {
        if (row < m && col < n) lds_c[row + m * col] = alpha * sum;
      });
      const bool has_in = d.cin_off >= 0;
      const int cbytes = m * n * 8;
      const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(c_out + d.c_off), 0, cbytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(c_in + (has_in ? d.cin_off : 0)), 0, has_in ? cbytes : 0, 0x00020000);
      typedef double f64x2 __attribute__((ext_vector_type(2)));
      constexpr int CC = (8 * TM * 8 * TN * 8 + 1023) / 1024;
      static_assert(CC * 1024 <= mid_lds_bytes(TM, TN, MID_KSL), "the C block must fit the wave's slice");
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        if (c * 1024 < cbytes) {   // (wave-uniform)
          f64x2 v = *reinterpret_cast<const f64x2*>(smem + c * 1024 + lane * 16);
          if (has_in) {
            const f64x2 w = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rsi, lane * 16, c * 1024, 0));
            v[0] += beta * w[0];
            v[1] += beta * w[1];
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsc, lane * 16 + c * 1024, 0, 2);
          const int idx = c * 128 + 2 * lane;
          if (idx < m * n) ss += v[0] * v[0];
          if (idx + 1 < m * n) ss += v[1] * v[1];
        }
      }
    } else {
      // beyond 32: C leaves straight from the accumulators, 8-byte stores, 4 rows x 8 columns of the block per instruction.  (Measured, session r06_14:
      // the epilogue above needs a 12.8 KB slice for 40 x 40 and was 3-5 % SLOWER at 20 products per C block: 40^3 8.5 against 8.2 ms.)
      double* C = c_out + d.c_off;
      const bool has_in = d.cin_off >= 0;
      const double* Ci = c_in + (has_in ? d.cin_off : 0);
      S.drain(lane, own_r, own_c, [&](int row, int col, double sum) {
        if (row < m && col < n) {
          double v = alpha * sum;
          if (has_in) v += beta * Ci[row + (size_t)m * col];
          C[row + (size_t)m * col] = v;
          ss += v * v;
        }
      });
    }
    // squared Frobenius norm of the block as it was stored: the final block filter of a filtered multiply reads it instead of C (as the exact-size kernels leave it)
    if (norms) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
      if (lane == 0) norms[cb_index] = ss;
    }
  }
}

}  // namespace dbcsr_amd
#endif
