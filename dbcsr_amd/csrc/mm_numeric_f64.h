// mm_numeric_f64.h -- fp64 block-product kernels: direct, LDS-staged, exact-size, pipelined, packed 4 x 4
// Part of the device-resident multiply engine: included by mm_engine.hip (one translation unit), in this order:
// mm_workspace.h, mm_symbolic.h, mm_numeric_f64.h, mm_numeric_f32.h, mm_aux.h.
#ifndef DBCSR_AMD_MM_NUMERIC_F64_H
#define DBCSR_AMD_MM_NUMERIC_F64_H

namespace dbcsr_amd {

// ----------------------------------------------------------------------------
// numeric kernels
// ----------------------------------------------------------------------------
template <int MA, int NC>
__device__ __forceinline__ void cblock_f64(const Desc& d, const Entry* __restrict__ entries, const double* __restrict__ a_data,
                                           const double* __restrict__ b_data, double* __restrict__ c_out,
                                           const double* __restrict__ c_in, double alpha, double beta, const LaneMap& L, int row0,
                                           int col0, double* __restrict__ norm_out = nullptr) {
  double acc[MA][NC];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = 0.0;
  const int m = d.m, n = d.n;
  const Entry* e = entries + d.prod_start;
  for (int p = 0; p < d.prod_cnt; ++p) {
    const uint64_t ao = e[p].a_off(), bo = e[p].b_off();
    block_product_f64<MA, NC, false>(acc, a_data + ao, b_data + bo, m, n, e[p].ks(), L, row0, col0);
  }
  double* C = c_out + d.c_off;
  const bool has_in = d.cin_off >= 0;
  const double* Ci = c_in + (has_in ? d.cin_off : 0);
  double ss = 0.0;
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int row = row0 + 8 * a + L.rowd, col = col0 + 8 * c + L.coll;
      if (row < m && col < n) {
        double v = alpha * acc[a][c];
        if (has_in) v += beta * Ci[row + (size_t)m * col];
        C[row + (size_t)m * col] = v;
        ss += v * v;
      }
    }
  // (one tile covers the whole block here: row0 = col0 = 0) the squared norm of the block as stored, for the final filter of a filtered multiply
  if (norm_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if ((threadIdx.x & 63) == 0) *norm_out = ss;
  }
}

__global__ void __launch_bounds__(256) mm_numeric_f64(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                      const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                      double* __restrict__ c_out, const double* __restrict__ c_in, double alpha,
                                                      double beta, int skip_empty) {
  const int lane = threadIdx.x & 63;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t cb = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6));
  if (cb >= nblk) return;
  const Desc d = descs[cb];
  if (skip_empty && d.prod_cnt == 0) return;
  const LaneMap L(lane);
  const int m = d.m, n = d.n;
  if (m <= 32 && n <= 32) {
    const int MA = (m + 7) >> 3, NC = (n + 7) >> 3;
    switch (MA * 4 + NC) {
#define DBCSR_CASE(A_, C_) \
  case A_ * 4 + C_: cblock_f64<A_, C_>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, L, 0, 0); break;
      DBCSR_CASE(1, 1) DBCSR_CASE(1, 2) DBCSR_CASE(1, 3) DBCSR_CASE(1, 4)
      DBCSR_CASE(2, 1) DBCSR_CASE(2, 2) DBCSR_CASE(2, 3) DBCSR_CASE(2, 4)
      DBCSR_CASE(3, 1) DBCSR_CASE(3, 2) DBCSR_CASE(3, 3) DBCSR_CASE(3, 4)
      DBCSR_CASE(4, 1) DBCSR_CASE(4, 2) DBCSR_CASE(4, 3) DBCSR_CASE(4, 4)
#undef DBCSR_CASE
      default: break;
    }
  } else {  // large blocks: 32 x 32 tiles, one after the other
    for (int row0 = 0; row0 < m; row0 += 32)
      for (int col0 = 0; col0 < n; col0 += 32) cblock_f64<4, 4>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, L, row0, col0);
  }
}

// ---- LDS-staged variant ----------------------------------------------------
// Measured on v1 (profiles/r01_v1_direct_loads_rocprofv3_summary.txt): loading
// MFMA fragments straight from global memory costs ~40 L1 accesses per wave
// load (TA 86 % busy, MFMA 20 % busy).  Here each wave copies the whole A and B
// block of a product with fully coalesced 16-byte loads into its private LDS
// slice (no barrier: one wave, in-order LDS queue) and reads fragments with
// ds_read_b64; the next product's blocks are already in flight in registers
// while the current one is multiplied.

template <int MA, int NC>
__device__ __forceinline__ void cblock_f64_lds(const Desc& d, const Entry* __restrict__ entries, const double* __restrict__ a_data,
                                               const double* __restrict__ b_data, double* __restrict__ c_out,
                                               const double* __restrict__ c_in, double alpha, double beta, const LaneMap& L, int lane,
                                               char* lds_a, char* lds_b, int dbg) {
  // Staging in 1 KiB chunks (64 lanes x 16 B).  The loads are RAW BUFFER loads whose
  // descriptor covers exactly one block: the hardware bounds check returns zeros past
  // the block end, which (a) needs no address arithmetic or tail fix-up on the VALU
  // (v3 profile: 284 VALU instructions per product against 54 MFMAs) and (b) zero-pads
  // A's k dimension in LDS for free.  Chunk counts are wave-uniform.
  constexpr int CA = 2 * MA, CB = 2 * NC;  // enough for (8 MA) x 32 and 32 x (8 NC) doubles
  double acc[MA][NC];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = 0.0;
  const int m = d.m, n = d.n;
  const Entry* e = entries + d.prod_start;
  const int cnt = d.prod_cnt;
  u32x4 ra[CA], rb[CB];
  const int voff = lane * 16;
  // The product-list entries are kept two ahead in scalar registers: entry p+1 is needed when product p's operands
  // have been copied to LDS (to start the next prefetch), so it is requested one trip earlier and its scalar-load
  // latency never sits between the LDS copy and the MFMAs.
  auto issue = [&](uint64_t a_off, uint64_t b_off_in, int ks) {
    // explicit scalarisation: with the k extent known to fit 16 bits the compiler multiplies on the VALU (mul24), the buffer
    // descriptor then sits in VGPRs and every load becomes a waterfall loop (measured: config 3 10.6 -> 12.7 ms)
    const int abytes = __builtin_amdgcn_readfirstlane(m * ks * 8), bbytes = __builtin_amdgcn_readfirstlane(ks * n * 8);
    const int nca = __builtin_amdgcn_readfirstlane((m * ((ks + 3) & ~3) * 8 + 1023) >> 10), ncb = (bbytes + 1023) >> 10;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + a_off), 0, abytes, 0x00020000);
    // dbg 128 (profiling only): fold all B blocks onto the first 1 MB of B -> L2-resident; isolates the cost of L2 misses
    const uint64_t b_off = (dbg & 128) ? (b_off_in % (uint64_t)(131072 - 1024)) : b_off_in;
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + b_off), 0, bbytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < CA; ++c)
      if (c < nca) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
    // (measured: a non-temporal hint (aux = 2) on these streamed B loads costs 20-30 % -- plain loads)
#pragma unroll
    for (int c = 0; c < CB; ++c)
      if (c < ncb) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
  };
  // dbg (ablation switches for profiling, 0 in production): 1 = no global loads, 2 = no MFMA/LDS reads, 4 = no LDS writes
  if (dbg & 1) {
#pragma unroll
    for (int c = 0; c < CA; ++c) ra[c] = u32x4{0u, 0x3ff00000u, 0u, 0x3ff00000u};
#pragma unroll
    for (int c = 0; c < CB; ++c) rb[c] = u32x4{0u, 0x3ff00000u, 0u, 0x3ff00000u};
  }
  Entry e0 = cnt > 0 ? e[0] : Entry::make(0, 0, 1);            // product p (being staged / multiplied)
  Entry e1 = cnt > 1 ? e[1] : e0;                            // product p + 1 (prefetched next)
  if (cnt > 0 && !(dbg & 1)) issue(e0.a_off(), e0.b_off(), e0.ks());
  for (int p = 0; p < cnt; ++p) {
    const int ks = e0.ks();
    if (!(dbg & 4)) {
      const int nca = __builtin_amdgcn_readfirstlane((m * ((ks + 3) & ~3) * 8 + 1023) >> 10), ncb = __builtin_amdgcn_readfirstlane((ks * n * 8 + 1023) >> 10);
#pragma unroll
      for (int c = 0; c < CA; ++c)
        if (c < nca) *reinterpret_cast<u32x4*>(lds_a + c * 1024 + voff) = ra[c];
      DBCSR_AMD_LDS_ORDER();
#pragma unroll
      for (int c = 0; c < CB; ++c)
        if (c < ncb) *reinterpret_cast<u32x4*>(lds_b + c * 1024 + voff) = rb[c];
    }
    if (p + 1 < cnt && !(dbg & 1)) issue(e1.a_off(), e1.b_off(), e1.ks());
    const Entry e2 = e[p + 2 < cnt ? p + 2 : cnt - 1];      // requested now, first used one trip later
    if (!(dbg & 2))
      block_product_f64_lds<MA, NC>(acc, reinterpret_cast<const double*>(lds_a), reinterpret_cast<const double*>(lds_b), m, n, ks, L);
    e0 = e1;
    e1 = e2;
  }
  double* C = c_out + d.c_off;
  const bool has_in = d.cin_off >= 0;
  const double* Ci = c_in + (has_in ? d.cin_off : 0);
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
      if (row < m && col < n) {
        double v = alpha * acc[a][c];
        if (has_in) v += beta * Ci[row + (size_t)m * col];
        C[row + (size_t)m * col] = v;  // plain store: non-temporal stores doubled WRITE_SIZE here (8-byte scattered lanes are not combined)
      }
    }
}

// ---- exact-size variant ------------------------------------------------------
// Specialisation for C blocks of M x N whose products have inner dimension K, all compile-time (what the
// reference's JIT does per (m, n, k) triple): chunk counts, LDS fragment offsets and the k loop are constants, so
// a product costs its buffer loads, LDS copies, ds_reads with immediate offsets and MFMAs and next to nothing else
// (the generic path: 103 VALU + 98 SALU instructions per 23^3 product besides the 54 MFMAs).  Products of the
// block with another inner dimension (the tail block column of A) are multiplied straight from global memory.
typedef const volatile double __attribute__((address_space(3))) lds_vd;  // volatile LDS read: never paired into ds_read2_b64
// VAR: 0 = production (no ablation branch is compiled in), 1 = the run-time ablation switches of DBCSR_AMD_MM_DBG (profiling),
// 2 = production with the fragment reads kept as single ds_read_b64 (the compiler pairs them into ds_read2_b64 otherwise),
// 3 / 4 = production + every product also touches one dword per cache line of the one / two A blocks stored before its own (the block
// row's neighbours): an A block is then referenced two / three times as often, against its eviction by the B stream
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// before_epilogue: called once, after the last product and before the C block is written (the persistent form of the kernel asks for
// its next position there: the round trip then runs under the epilogue)
template <int M, int N, int K, int VAR, class Hook = NoHook>
__device__ __forceinline__ void cblock_f64_exact(const Desc& d, const Entry first, const Entry* __restrict__ entries, const double* __restrict__ a_data,
                                                 const double* __restrict__ b_data, double* __restrict__ c_out,
                                                 const double* __restrict__ c_in, double alpha, double beta, const LaneMap& L, int lane,
                                                 char* lds_a, char* lds_b, int dbg_rt, double* __restrict__ norm_out, Hook before_epilogue = Hook(),
                                                 double drop_below = 0.0) {
  const int dbg = VAR == 1 ? dbg_rt : 0;
  constexpr int MA = (M + 7) / 8, NC = (N + 7) / 8, KS = (K + 3) / 4, K4 = 4 * KS;
  constexpr int CA = (M * K4 * 8 + 1023) / 1024, CB = (K * N * 8 + 1023) / 1024;
  // Columns of 16 or 32 doubles are staged with a pitch of + 2 doubles (round 6; the run-time compiled class kernels have had it since round 2,
  // mm_exact.h: Pitch): with a column stride of 128 or 256 bytes the 8 columns of a B fragment read -- and the k index of an A fragment read -- all
  // fall into the same LDS banks.  Measured on 32^3 blocks (profiles/r06_slab_kernel.txt): hot<32,32,32> 25.0 ms where hot<30,30,30> scales to 15.1.
  // (B columns of 24 doubles too: a stride of 192 bytes leaves 4 distinct bank groups for 8 columns, 2-way conflicts)
  // (23 x 23: columns of 46 dwords leave one pair of banks shared by two of the eight columns of a B fragment read; reading at a conflict-free pitch of
  // 25 -- timing only, session r06_20 -- gave 19.20 -> 19.04 ms on config 2: not worth staging B in 8-byte granules)
  constexpr int APAD = (M % 16 == 0) ? 2 : 0, BPAD = (K % 8 == 0) ? 2 : 0, AP = M + APAD, BP = K + BPAD;
  static_assert(APAD == 0 || M % 2 == 0, "a lane's 16-byte granule (two doubles) must not straddle two padded columns");
  static_assert(BPAD == 0 || K % 2 == 0, "a lane's 16-byte granule (two doubles) must not straddle two padded columns");
  double acc[MA][NC];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = 0.0;
  const Entry* e = entries + d.prod_start;
  const int cnt = d.prod_cnt;
  u32x4 ra[CA], rb[CB];
  const int voff = lane * 16;
  // byte offset in the staged image of the 16-byte granule this lane carries of piece c: (with padding) 16 bytes more per column before it
  auto soff_a = [&](int c) { return c * 1024 + voff + (APAD ? 8 * APAD * ((c * 128 + lane * 2) / M) : 0); };
  auto soff_b = [&](int c) { return c * 1024 + voff + (BPAD ? 8 * BPAD * ((c * 128 + lane * 2) / K) : 0); };
  unsigned touch = 0;  // VAR 3 / 4: the dwords of the keep-alive loads, folded so that each is waited for one product later
  // fragment addresses: constant for the whole life of the wave
  const double* pa[MA];
  const double* pb[NC];
  const double* pbt[NC];  // last k step when K is not a multiple of 4: lanes past the end read element (0, col) (A's padding is zero)
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    int row = 8 * a + L.rowl;
    row = row < M ? row : M - 1;
    pa[a] = reinterpret_cast<const double*>(lds_a) + row + AP * L.kq;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = 8 * c + L.coll;
    col = col < N ? col : N - 1;
    pb[c] = reinterpret_cast<const double*>(lds_b) + L.kq + BP * col;
    const int kt = 4 * (KS - 1) + L.kq;
    pbt[c] = reinterpret_cast<const double*>(lds_b) + (kt < K ? kt : 0) + BP * col;
  }
  auto issue = [&](uint64_t a_off, uint64_t b_off_in) {
    if (dbg & 1) return;
    const uint32_t fold = (dbg >> 16) ? (uint32_t)(dbg >> 16) * 65536u : 131072u;  // B window of the L2/MALL experiments, doubles
    const uint64_t b_off = (dbg & 128) ? (b_off_in % (uint64_t)(fold - 1024u)) : b_off_in;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + a_off), 0, M * K * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + b_off), 0, K * N * 8, 0x00020000);
#pragma unroll
    for (int c = 0; c < CA; ++c) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
    // (measured on the streamed B loads: sc0 / sc1 / sc0+sc1 make no difference, nt costs +30 %)
#pragma unroll
    for (int c = 0; c < CB; ++c) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
    if constexpr (VAR == 3 || VAR == 4) {
      constexpr uint64_t BLK = (uint64_t)M * K;
      constexpr int NB = VAR == 3 ? 1 : 2;
      const uint64_t back = a_off >= NB * BLK ? NB * BLK : 0;  // (wave-uniform; the first blocks of A touch themselves)
      const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + a_off - back), 0, (int)(NB * BLK * 8), 0x00020000);
      touch ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rst, lane * 128, 0, 0);
      if constexpr (NB * BLK * 8 > 64 * 128) touch ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rst, lane * 128, 64 * 128, 0);
    }
  };
  // Products with inner dimension K run through the staged pipeline (i0 = the one being multiplied, i1 = the next
  // candidate, whose list entry was requested one trip earlier); the others are summed afterwards.
  int i0 = 0;
  Entry e0 = first;  // == e[0], already here
  while (i0 < cnt && e0.ks() != K) {
    ++i0;
    e0 = e[i0 < cnt ? i0 : cnt - 1];
  }
  int i1 = i0 + 1;
  Entry e1 = e[i1 < cnt ? i1 : cnt - 1];
  if (dbg & 1) {
#pragma unroll
    for (int c = 0; c < CA; ++c) ra[c] = u32x4{0u, 0x3ff00000u, 0u, 0x3ff00000u};
#pragma unroll
    for (int c = 0; c < CB; ++c) rb[c] = u32x4{0u, 0x3ff00000u, 0u, 0x3ff00000u};
  }
  if (i0 < cnt) issue(e0.a_off(), e0.b_off());
  while (i0 < cnt) {
    if (!(dbg & 4)) {
#pragma unroll
      for (int c = 0; c < CA; ++c) *reinterpret_cast<u32x4*>(lds_a + soff_a(c)) = ra[c];
      DBCSR_AMD_LDS_ORDER();
#pragma unroll
      for (int c = 0; c < CB; ++c) *reinterpret_cast<u32x4*>(lds_b + soff_b(c)) = rb[c];
    }
    while (i1 < cnt && e1.ks() != K) {
      ++i1;
      e1 = e[i1 < cnt ? i1 : cnt - 1];
    }
    if (i1 < cnt) issue(e1.a_off(), e1.b_off());
    const Entry e2 = e[i1 + 1 < cnt ? i1 + 1 : cnt - 1];
    // VAR 5 / 6 (round 5): the wave raises its issue priority for the burst of MFMAs (and their fragment reads) of a
    // product, so that the SIMD's other waves' copies and address arithmetic do not sit between two of its MFMAs; 6 = 5 + unpaired reads
    if constexpr (VAR == 5 || VAR == 6) __builtin_amdgcn_s_setprio(2);
    if (!(dbg & 2))
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      double av[MA], bv[NC];
#pragma unroll
      for (int a = 0; a < MA; ++a) {
        if constexpr (VAR == 2 || VAR == 6)
          av[a] = *(lds_vd*)(pa[a] + s * 4 * AP);
        else
          av[a] = pa[a][s * 4 * AP];
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if constexpr (VAR == 2 || VAR == 6)
          bv[c] = (s == KS - 1 && (K & 3)) ? *(lds_vd*)(pbt[c]) : *(lds_vd*)(pb[c] + 4 * s);
        else
          bv[c] = (s == KS - 1 && (K & 3)) ? pbt[c][0] : pb[c][4 * s];
      }
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
    }
    if constexpr (VAR == 5 || VAR == 6) __builtin_amdgcn_s_setprio(0);
    i0 = i1;
    e0 = e1;
    i1 = i1 + 1;
    e1 = e2;
  }
  for (int p = 0; p < cnt; ++p) {
    const Entry ep = e[p];
    if (ep.ks() != K) block_product_f64<MA, NC, false>(acc, a_data + ep.a_off(), b_data + ep.b_off(), M, N, ep.ks(), L);
  }
  if constexpr (VAR == 3 || VAR == 4) asm volatile("" ::"v"(touch));  // the keep-alive loads are loads the compiler must keep
  before_epilogue();
  const bool has_in = d.cin_off >= 0;
  // A filtered multiply whose final block filter is known (dbcsr_amd_mm_expect_filter; drop_below = its eps^2): a NEW block (no C_in: the usual case of a sparse
  // product) has its norm in the accumulators -- formed here, before anything touches LDS, and a block the filter is going to drop (||blk||^2 < eps^2: the double
  // written to norm_out, the comparison of filter_flags) is neither staged nor written.  Nobody reads it.  (Blocks with C_in take the epilogue below as always.)
  if (norm_out && drop_below > 0.0 && !has_in && !(dbg & 8)) {
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
  if (dbg & 8) {  // scattered 8-byte stores straight from the accumulators (the first version; kept for comparison)
    double* C = c_out + d.c_off;
    const double* Ci = c_in + (has_in ? d.cin_off : 0);
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
        if (row < M && col < N) {
          double v = alpha * acc[a][c];
          if (has_in) v += beta * Ci[row + (size_t)M * col];
          C[row + (size_t)M * col] = v;
        }
      }
    return;
  }
  // C epilogue through LDS: the block is laid out as stored (column-major, contiguous) in the wave's staging area and
  // leaves in whole 1 KiB pieces -- 16 B per lane, full cache lines except at the two ends of the block -- with the
  // streaming hint, so that the 8.6 GB of C that config 2 writes do not push the A block-rows out of L2 / the B panel out
  // of the Infinity Cache.  (Non-temporal on the scattered 8-byte stores doubled WRITE_SIZE: partial lines are not combined.)
  constexpr int CC = (M * N * 8 + 1023) / 1024;
  double* lds_c = reinterpret_cast<double*>(lds_a);
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
      if (row < M && col < N) lds_c[row + M * col] = alpha * acc[a][c];
    }
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(c_out + d.c_off), 0, M * N * 8, 0x00020000);
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  double ss = 0.0;  // squared Frobenius norm of the block as it is stored (filtered multiplies): summed as the values leave
  // The 16-byte stores below carry their piece offset in the VECTOR / immediate offset, never in the scalar offset: a buffer store of more than 64 bits whose soffset is an
  // SGPR is NOT covered by the compiler's store-data hazard rule (it assumes none), yet on gfx950 a VALU write to the data registers right behind such a store reaches the
  // store: the class (9, 32) kernel returned 16 elements per block with the low dword 0x100 (the next instruction's constant) in 0.2 % of the blocks (round 6, session 56).
  if (has_in) {
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(c_in + d.cin_off), 0, M * N * 8, 0x00020000);
    u32x4 ci[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) ci[c] = __builtin_amdgcn_raw_buffer_load_b128(rsi, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      f64x2 v = *reinterpret_cast<const f64x2*>(lds_a + c * 1024 + voff);
      const f64x2 w = __builtin_bit_cast(f64x2, ci[c]);
      v[0] += beta * w[0];
      v[1] += beta * w[1];
      if (norm_out) {  // (the final values go into the norm as they leave: no second pass over the slice)
        const int idx = c * 128 + 2 * lane;
        if (idx < M * N) ss += v[0] * v[0];
        if (idx + 1 < M * N) ss += v[1] * v[1];
      }
      if (dbg & 16)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsc, voff + c * 1024, 0, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsc, voff + c * 1024, 0, 2);
    }
  } else {
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds_a + c * 1024 + voff);
      if (norm_out) {
        const f64x2 x = __builtin_bit_cast(f64x2, v);
        const int idx = c * 128 + 2 * lane;
        if (idx < M * N) ss += x[0] * x[0];
        if (idx + 1 < M * N) ss += x[1] * x[1];
      }
      if (dbg & 16)
        __builtin_amdgcn_raw_buffer_store_b128(v, rsc, voff + c * 1024, 0, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b128(v, rsc, voff + c * 1024, 0, 2);
    }
  }
  // squared Frobenius norm of the block as it was stored (the final block filter of a filtered multiply reads it instead of C):
  // the block still sits in the wave's LDS slice
  if (norm_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if (lane == 0) *norm_out = ss;
  }
}

// C blocks of exactly M x N take the exact-size path; every other block of the launch the generic one.
template <int M, int N, int K, int VAR>
__global__ void __launch_bounds__(256) mm_numeric_f64_hot(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                          const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                          double* __restrict__ c_out, const double* __restrict__ c_in, double alpha,
                                                          double beta, int lds_a_doubles, int lds_wave_doubles, int dbg, const int* __restrict__ order,
                                                          const Work* __restrict__ work, double* __restrict__ norms) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t pos = (int64_t)wg * (int)(blockDim.x >> 6) + wid;  // 1, 2 or 4 waves per workgroup (Engine::wg_waves)
  // launch-order records (build_work): descriptor and first product in one read -- no order[] -> descs[] -> entries[] chain
  const Work w = work[pos];
  if (w.prod_cnt < 0) return;  // padding position
  // norms[nblk]: the threshold of the final block filter when the host announced it (0: every block is written); asked for now, needed in the epilogue
  const double drop_below = norms ? norms[nblk] : 0.0;
  const Desc d = {w.c_off, w.cin_off, w.prod_start, w.prod_cnt, w.m, w.n};
  Entry first;
  first.a_lo = w.a_lo, first.b_lo = w.b_lo, first.w = w.w;
  if ((dbg & 32) && d.prod_cnt == 0) return;
  if ((dbg & 64) && d.m == M && d.n == N) return;  // the tile kernel (mm_tile.h) computed the blocks of the dominant size
  // profiling variant only: bits 8-15 of DBCSR_AMD_MM_DBG switch whole XCDs off (their C blocks are simply not computed): how fast is an
  // XCD whose neighbours leave the fabric alone?
  if constexpr (VAR == 1)
    if ((dbg >> 8) & 0xff & (1 << (blockIdx.x & 7))) return;
  char* lds_a = smem + (size_t)wid * lds_wave_doubles * 8;
  char* lds_b = lds_a + (size_t)lds_a_doubles * 8;
  const LaneMap L(lane);
  if (d.m == M && d.n == N) {
    cblock_f64_exact<M, N, K, VAR>(d, first, entries, a_data, b_data, c_out, c_in, alpha, beta, L, lane, lds_a, lds_b, dbg,
                              norms ? norms + w.cb : nullptr, NoHook(), drop_below);
    return;
  }
  // the few blocks of another size (tail block row / column): straight from global memory, as one 32 x 32 tile (they leave their norm too: no pass over
  // all descriptors afterwards to find them -- 0.98 ms of a filtered multiply on config 4's 14 M C blocks, session r06_55)
  cblock_f64<4, 4>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, L, 0, 0, norms ? norms + w.cb : nullptr);
}

// PERSISTENT form of the kernel above (DBCSR_AMD_MM_HOT_PERSISTENT=1; an experiment, and the groundwork of a launch that leaves some
// XCDs to another kernel): as many one-wave workgroups as the chip holds at once; the waves of XCD x take the positions of
// x's stream of the launch order one after the other from a counter of their own (an L2 atomic inside the XCD), so the frontier of
// the stream stays as compact as under the hardware dispatcher -- unlike a static share of positions per wave, which drifts apart
// (section 7, "STREAM kernel").  XCDs outside `xcd_mask` leave at once and their streams stay untouched.
template <int M, int N, int K>
__global__ void __launch_bounds__(64) mm_numeric_f64_hot_persistent(const Entry* __restrict__ entries, const double* __restrict__ a_data,
                                                                    const double* __restrict__ b_data, double* __restrict__ c_out,
                                                                    const double* __restrict__ c_in, double alpha, double beta, int lds_a_doubles,
                                                                    int dbg, const Work* __restrict__ work, long stream_len,
                                                                    unsigned* __restrict__ counters, unsigned xcd_mask, double* __restrict__ norms) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const unsigned xcd = blockIdx.x & 7u;
  if (!((xcd_mask >> xcd) & 1u)) return;
  char* lds_a = smem;
  char* lds_b = lds_a + (size_t)lds_a_doubles * 8;
  const LaneMap L(lane);
  const Work* stream = work + (long)xcd * stream_len;
  // ONE position per request and nothing taken ahead: what is in flight on the XCD must stay a window of ~512 consecutive positions
  // (1.2 block rows of A: its L2 share), as under the hardware dispatcher.  Measured on config 2 (profiles/r03_hot_persistent.txt):
  // positions taken two ahead or in chunks of eight widen the window to several block rows -- 230 GB over the fabric instead of 145,
  // L2 hit rate 0.16 instead of 0.47, 29-31 ms; one request per block with its round trip, the record and the first operands all
  // in a row between two blocks: 27 ms.  So the request goes out right before the C block is written and returns under the epilogue.
  // The counter of an XCD has a cache line of its own and is incremented with WORKGROUP scope: the atomic is then carried out in this
  // XCD's L2 -- every wave that uses it runs on this XCD -- instead of going out to memory as a device-scope atomic does (eight
  // counters in one line, device scope: 2 M fabric round trips serialised on one line, 27 ms whatever else was hidden).
  auto take = [&]() {
    unsigned v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add(&counters[32 * xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return v;  // (lane 0's value; made uniform where it is used)
  };
  unsigned pos = (unsigned)__builtin_amdgcn_readfirstlane((int)take());
  while ((long)pos < stream_len) {
    const Work w = stream[pos];
    unsigned next_raw = 0;
    if (w.prod_cnt >= 0 && !((dbg & 32) && w.prod_cnt == 0) && w.m == M && w.n == N) {
      const Desc d = {w.c_off, w.cin_off, w.prod_start, w.prod_cnt, w.m, w.n};
      Entry first;
      first.a_lo = w.a_lo, first.b_lo = w.b_lo, first.w = w.w;
      auto hook = [&]() { next_raw = take(); };
      cblock_f64_exact<M, N, K, 0, decltype(hook)>(d, first, entries, a_data, b_data, c_out, c_in, alpha, beta, L, lane, lds_a, lds_b, 0,
                                                   norms ? norms + w.cb : nullptr, hook);
    } else {
      next_raw = take();
      if (w.prod_cnt >= 0 && !((dbg & 32) && w.prod_cnt == 0)) {
        const Desc d = {w.c_off, w.cin_off, w.prod_start, w.prod_cnt, w.m, w.n};
        cblock_f64<4, 4>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, L, 0, 0);
      }
    }
    pos = (unsigned)__builtin_amdgcn_readfirstlane((int)next_raw);
  }
}

// all block dimensions of the launch are <= 8*MAXT (<= 32); lds_wave_doubles = per-wave LDS slice (A part then B part).
// MAXT bounds the register allocation to what the largest block class present needs.
template <int MAXT>
__global__ void __launch_bounds__(256) mm_numeric_f64_lds(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                          const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                          double* __restrict__ c_out, const double* __restrict__ c_in, double alpha,
                                                          double beta, int lds_a_doubles, int lds_wave_doubles, int dbg, const int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t pos = (int64_t)wg * (int)(blockDim.x >> 6) + wid;  // gridDim.x * waves per workgroup == padded length of order[]
  const int64_t cb = order[pos];
  if (cb < 0 || cb >= nblk) return;
  if ((dbg & 32) && descs[cb].prod_cnt == 0) return;  // in-place accumulation (beta = 1): untouched blocks stay as they are
  char* lds_a = smem + (size_t)wid * lds_wave_doubles * 8;
  char* lds_b = lds_a + (size_t)lds_a_doubles * 8;
  const Desc d = descs[cb];
  const LaneMap L(lane);
  const int MA = (d.m + 7) >> 3, NC = (d.n + 7) >> 3;
  switch (MA * 4 + NC) {
#define DBCSR_CASE(A_, C_)                                                                                           \
  case A_ * 4 + C_:                                                                                                  \
    if constexpr (A_ <= MAXT && C_ <= MAXT)                                                                          \
      cblock_f64_lds<A_, C_>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, L, lane, lds_a, lds_b, dbg);          \
    break;
    DBCSR_CASE(1, 1) DBCSR_CASE(1, 2) DBCSR_CASE(1, 3) DBCSR_CASE(1, 4)
    DBCSR_CASE(2, 1) DBCSR_CASE(2, 2) DBCSR_CASE(2, 3) DBCSR_CASE(2, 4)
    DBCSR_CASE(3, 1) DBCSR_CASE(3, 2) DBCSR_CASE(3, 3) DBCSR_CASE(3, 4)
    DBCSR_CASE(4, 1) DBCSR_CASE(4, 2) DBCSR_CASE(4, 3) DBCSR_CASE(4, 4)
#undef DBCSR_CASE
    default: break;
  }
}


// ---- pipelined variant: a wave walks a RANGE of C blocks --------------------------
// v4 measurements: with one C block per wave, every wave pays the dependent chain
// order -> descriptor -> entry -> operand loads -> LDS before its first MFMA (about 5 us);
// at 14 products per block that is ~10 % of a wave's life, at 1-4 products per block
// (configs 3 and 4) it dominates.  Here a wave owns G consecutive positions of order[] and
// the product pipeline runs ACROSS C-block boundaries: while the last product of block b is
// multiplied, the first product of block b+1 is already in flight, and the descriptor of
// block b+2 has been requested.
struct PipeCtx {
  const Desc* __restrict__ descs;
  const Entry* __restrict__ entries;
  const double* __restrict__ a_data;
  const double* __restrict__ b_data;
  double* __restrict__ c_out;
  const double* __restrict__ c_in;
  double alpha, beta;
  char* lds_a;
  char* lds_b;
  int lane, voff;
};

__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// descriptor held in scalar registers
__device__ __forceinline__ Desc load_desc_uniform(const Desc* __restrict__ descs, int cb) {
  const Desc t = descs[cb];
  Desc d;
  d.c_off = uniform64(t.c_off);
  d.cin_off = uniform64(t.cin_off);
  d.prod_start = uniform64(t.prod_start);
  d.prod_cnt = __builtin_amdgcn_readfirstlane(t.prod_cnt);
  const int mn = __builtin_amdgcn_readfirstlane(((int)(uint16_t)t.m) | (((int)(uint16_t)t.n) << 16));
  d.m = (int16_t)(mn & 0xffff);
  d.n = (int16_t)(mn >> 16);
  return d;
}

// prefetch product `pidx` (index into entries) of a C block of size m x n into the staging registers
template <int CMAX>
__device__ __forceinline__ void pipe_issue(const PipeCtx& X, int64_t pidx, int m, int n, u32x4 (&ra)[CMAX], u32x4 (&rb)[CMAX]) {
  const Entry e = X.entries[pidx];
  Entry u;  // wave-uniform copy
  u.a_lo = __builtin_amdgcn_readfirstlane(e.a_lo);
  u.b_lo = __builtin_amdgcn_readfirstlane(e.b_lo);
  u.w = __builtin_amdgcn_readfirstlane(e.w);
  const uint64_t ao = u.a_off(), bo = u.b_off();
  const int ks = u.ks();
  const int abytes = __builtin_amdgcn_readfirstlane(m * ks * 8), bbytes = __builtin_amdgcn_readfirstlane(ks * n * 8);  // scalar on purpose, see cblock_f64_lds
  const int nca = __builtin_amdgcn_readfirstlane((m * ((ks + 3) & ~3) * 8 + 1023) >> 10), ncb = (bbytes + 1023) >> 10;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(X.a_data + ao), 0, abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(X.b_data + bo), 0, bbytes, 0x00020000);
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < nca) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, X.voff, c * 1024, 0);
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < ncb) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, X.voff, c * 1024, 0);
}

// multiply the staged product into the accumulators of the current block.  acc is the launch-wide
// [MAXT][MAXT] array; class (MA, NC) uses a corner of it.
template <int MA, int NC, int MAXT>
__device__ __forceinline__ void pipe_compute(const PipeCtx& X, int m, int n, int ks, double (&acc)[MAXT][MAXT], const LaneMap& L) {
  double t[MA][NC];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) t[a][c] = acc[a][c];
  block_product_f64_lds<MA, NC>(t, reinterpret_cast<const double*>(X.lds_a), reinterpret_cast<const double*>(X.lds_b), m, n, ks, L);
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = t[a][c];
}

// write a finished block (any class: rows/columns outside the block are masked) and clear the accumulators
template <int MAXT>
__device__ __forceinline__ void pipe_flush(const PipeCtx& X, const Desc& d, double (&acc)[MAXT][MAXT], const LaneMap& L) {
  const int m = d.m, n = d.n;
  double* C = X.c_out + d.c_off;
  const bool has_in = d.cin_off >= 0;
  const double* Ci = X.c_in + (has_in ? d.cin_off : 0);
#pragma unroll
  for (int a = 0; a < MAXT; ++a)
#pragma unroll
    for (int c = 0; c < MAXT; ++c) {
      const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
      if (row < m && col < n) {
        double v = X.alpha * acc[a][c];
        if (has_in) v += X.beta * Ci[row + (size_t)m * col];
        C[row + (size_t)m * col] = v;
      }
      acc[a][c] = 0.0;
    }
}

template <int MAXT>
__global__ void __launch_bounds__(256) mm_numeric_f64_pipe(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                           const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                           double* __restrict__ c_out, const double* __restrict__ c_in, double alpha,
                                                           double beta, int lds_a_doubles, int lds_wave_doubles, int skip_empty,
                                                           const int* __restrict__ order, int64_t npos, int G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CMAX = 2 * MAXT;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  int64_t pos = ((int64_t)wg * 4 + wid) * G;
  const int64_t pos_end = min(pos + G, npos);
  if (pos >= npos) return;
  PipeCtx X;
  X.descs = descs; X.entries = entries; X.a_data = a_data; X.b_data = b_data; X.c_out = c_out; X.c_in = c_in;
  X.alpha = alpha; X.beta = beta;
  X.lds_a = smem + (size_t)wid * lds_wave_doubles * 8;
  X.lds_b = X.lds_a + (size_t)lds_a_doubles * 8;
  X.lane = lane; X.voff = lane * 16;
  const LaneMap L(lane);
  // Next C block of this wave's range that has products.  Blocks without products are finished on the spot
  // (C = beta*C_in or 0), or left untouched when accumulating in place (skip_empty).
  auto next_block = [&](Desc& d) -> bool {
    while (pos < pos_end) {
      const int cb = __builtin_amdgcn_readfirstlane(order[pos]);
      ++pos;
      if (cb < 0 || cb >= nblk) continue;
      d = load_desc_uniform(descs, cb);
      if (d.prod_cnt > 0) return true;
      if (!skip_empty) {
        double* C = c_out + d.c_off;
        const int ne = (int)d.m * (int)d.n;
        if (d.cin_off >= 0) {
          const double* Ci = c_in + d.cin_off;
          for (int e = lane; e < ne; e += 64) C[e] = beta * Ci[e];
        } else {
          for (int e = lane; e < ne; e += 64) C[e] = 0.0;
        }
      }
    }
    return false;
  };
  u32x4 ra[CMAX], rb[CMAX];
  double acc[MAXT][MAXT];
#pragma unroll
  for (int a = 0; a < MAXT; ++a)
#pragma unroll
    for (int c = 0; c < MAXT; ++c) acc[a][c] = 0.0;
  Desc cur, nxt, done;
  if (!next_block(cur)) return;
  bool have_nxt = next_block(nxt);
  bool pending = false;  // `done` is finished and still sits in acc, waiting to be written
  // Flat product loop.  p = -1: nothing staged yet (the first trip only issues the first prefetch), so there is
  // exactly ONE prefetch site and ONE LDS-write site in the kernel (one set of staging registers).
  // Order inside a trip: [wait for the prefetched operands, copy them to LDS] [write out the block finished in
  // the previous trip] [prefetch] [multiply].  The finished block's stores are issued BEFORE the next prefetch,
  // so the in-order vmcnt wait of the following trip never has to drain stores that were issued after loads.
  int p = -1, ks = 0;
  for (;;) {
    if (p >= 0) {
      ks = __builtin_amdgcn_readfirstlane(entries[cur.prod_start + p].ks());
      const int nca = __builtin_amdgcn_readfirstlane((cur.m * ((ks + 3) & ~3) * 8 + 1023) >> 10), ncb = __builtin_amdgcn_readfirstlane((ks * cur.n * 8 + 1023) >> 10);
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < nca) *reinterpret_cast<u32x4*>(X.lds_a + c * 1024 + X.voff) = ra[c];
      DBCSR_AMD_LDS_ORDER();
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < ncb) *reinterpret_cast<u32x4*>(X.lds_b + c * 1024 + X.voff) = rb[c];
    }
    if (pending) {
      pipe_flush<MAXT>(X, done, acc, L);
      pending = false;
    }
    const bool more = p + 1 < cur.prod_cnt;
    if (more || have_nxt) pipe_issue<CMAX>(X, more ? cur.prod_start + p + 1 : nxt.prod_start, more ? cur.m : nxt.m, more ? cur.n : nxt.n, ra, rb);
    if (p >= 0) {
      const int MA = (cur.m + 7) >> 3, NC = (cur.n + 7) >> 3;
      switch (MA * 4 + NC) {
#define DBCSR_CASE(A_, C_)                                                                               \
  case A_ * 4 + C_:                                                                                      \
    if constexpr (A_ <= MAXT && C_ <= MAXT) pipe_compute<A_, C_, MAXT>(X, cur.m, cur.n, ks, acc, L);     \
    break;
        DBCSR_CASE(1, 1) DBCSR_CASE(1, 2) DBCSR_CASE(1, 3) DBCSR_CASE(1, 4)
        DBCSR_CASE(2, 1) DBCSR_CASE(2, 2) DBCSR_CASE(2, 3) DBCSR_CASE(2, 4)
        DBCSR_CASE(3, 1) DBCSR_CASE(3, 2) DBCSR_CASE(3, 3) DBCSR_CASE(3, 4)
        DBCSR_CASE(4, 1) DBCSR_CASE(4, 2) DBCSR_CASE(4, 3) DBCSR_CASE(4, 4)
#undef DBCSR_CASE
        default: break;
      }
      if (!more) {  // block complete: it is written at the start of the next trip (or after the loop)
        done = cur;
        pending = true;
        if (!have_nxt) break;
        cur = nxt;
        have_nxt = next_block(nxt);
        p = 0;
        continue;
      }
    }
    ++p;
  }
  if (pending) pipe_flush<MAXT>(X, done, acc, L);
}

// The block-size statistics of a multiply in ONE launch (three workgroups: C's rows, the inner dimension, C's columns) instead of twelve -- max_of, mode_of,
// units_mode_of, size_hist per dimension and a memset --, each a few microseconds of work behind a launch of its own on the path of a multiply whose plan is
// not reused (round 6, session r06_52: 40 launches in front of the product kernel of config 1, 0.35 of its 0.86 ms).  Same results in the same places:
// mx[2 d] = max, mx[2 d + 1] = -min; md[2 d] = most frequent size in 1 ... 32 (ties: the smallest), md[2 d + 1] = how often; um[0 .. 3] = most frequent size in
// units of 4 (sizes 1 ... 48; ties: the largest) of C's rows and of its columns and how often; hist (may be null): 33 bins per dimension in the order rows,
// columns, inner (bin 0: sizes outside 1 ... 32).
__global__ void __launch_bounds__(256) block_size_stats(const int* __restrict__ rows, int nbr, const int* __restrict__ inner, int nbk,
                                                        const int* __restrict__ cols, int nbc, int* __restrict__ mx, int* __restrict__ md,
                                                        int* __restrict__ um, int* __restrict__ hist) {
  __shared__ int h[33], hu[13], red[8];
  const int d = blockIdx.x;  // 0: rows (m), 1: inner (k), 2: columns (n)
  const int* v = d == 0 ? rows : (d == 1 ? inner : cols);
  const int n = d == 0 ? nbr : (d == 1 ? nbk : nbc);
  if (threadIdx.x < 33) h[threadIdx.x] = 0;
  if (threadIdx.x < 13) hu[threadIdx.x] = 0;
  __syncthreads();
  int vmax = 0, vmin = -0x7fffffff;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int s = v[i];
    vmax = max(vmax, s);
    vmin = max(vmin, -s);
    atomicAdd(&h[(s >= 1 && s <= 32) ? s : 0], 1);
    if (s >= 1 && s <= 48) atomicAdd(&hu[(s + 3) >> 2], 1);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    vmax = max(vmax, __shfl_down(vmax, off, 64));
    vmin = max(vmin, __shfl_down(vmin, off, 64));
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax, red[4 + (threadIdx.x >> 6)] = vmin;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx[2 * d] = max(max(red[0], red[1]), max(red[2], red[3]));
    mx[2 * d + 1] = max(max(red[4], red[5]), max(red[6], red[7]));
    int best = 0, cnt = 0;
    for (int s = 1; s <= 32; ++s)
      if (h[s] > cnt) cnt = h[s], best = s;
    md[2 * d] = best;
    md[2 * d + 1] = cnt;
    if (d != 1) {
      best = 0, cnt = 0;
      for (int u = 12; u >= 1; --u)
        if (hu[u] > cnt) cnt = hu[u], best = u;
      um[d] = best;        // (d = 0: um[0 .. 1], d = 2: um[2 .. 3])
      um[d + 1] = cnt;
    }
  }
  if (hist && threadIdx.x < 33) hist[(d == 0 ? 0 : (d == 2 ? 33 : 66)) + threadIdx.x] = h[threadIdx.x];
}

// ---- fp64, C blocks of at most 4 x 4 (BASELINE config 1: 4 x 4 x 4 blocks) -------------------------------------------
// v_mfma_f64_4x4x4_4b_f64 multiplies FOUR independent 4x4x4 block triples at once (lane bits 2-3 select the triple).  With
// one wave per C block three quarters of every instruction are padding and a wave lives for ten products; here a wave
// owns four C blocks, one per MFMA sub-block, each walking its own product list, and every lane fetches exactly the A and
// B element it feeds (no LDS, no staging): per block product 2 element loads per lane and one MFMA per 4 of k.
// Sub-block b of lane l: b = (l >> 2) & 3; operands A[i = l & 3][k = l >> 4], B[k = l >> 4][j = l & 3]; result C[i = l >> 4][j = l & 3].
//
// The kernel is a chain of memory round trips (order -> descriptor -> entry -> elements), not arithmetic: what sets its time is how
// many of them a wave puts in flight at once.  The 16 lanes of a sub-block therefore read 16 ENTRIES of their list in one request
// (192 contiguous bytes), hand them round with ds_bpermute, and request the elements of eight products (16 loads per lane) before the
// first MFMA: a list of ten products costs 1 + 2 round trips instead of 11.  K4: every k extent is at most 4 (one MFMA per product).
// Config 1 (4096^2, 10.8 M products): 0.53 -> 0.30 ms.  Measured and not kept (docs/LOG_r04.md, sessions 14-16): 16 products per batch
// (0.37: two waves per SIMD fewer), persistent waves that pipeline order / descriptor / entries / elements of four successive quads into
// one round trip per quad (0.33), B column panels of 1-4 MB for the XCD's L2 (-3 %).  At 0.30 ms the launch moves 1.43 GB over the
// fabric (4.7 TB/s, 0.6 of its ceiling) in 33 M 128-byte L2 requests: a gather of 256 operand bytes per 128 flop, nothing to reuse.
template <bool K4>
__global__ void __launch_bounds__(256) mm_numeric_f64_tiny(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                           const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                           double* __restrict__ c_out, const double* __restrict__ c_in, double alpha,
                                                           double beta, int skip_empty, const int* __restrict__ order) {
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int sub = (lane >> 2) & 3, x = lane & 3, kq = lane >> 4;
  const int64_t pos = ((int64_t)wg * 4 + wid) * 4 + sub;  // gridDim.x * 16 == padded length of order[]
  const int cb = order[pos];
  const bool live = cb >= 0 && cb < nblk;
  Desc d;
  d.prod_cnt = 0;
  d.m = d.n = 0;
  d.c_off = d.cin_off = d.prod_start = 0;
  if (live) d = descs[cb];
  const int m = d.m, n = d.n;
  const Entry* e = entries + d.prod_start;
  const int cnt = live ? d.prod_cnt : 0;
  const int t = x + 4 * kq;  // this lane's place among the 16 lanes of its sub-block
  // C_in is requested now, ahead of the whole product walk
  double cin = 0.0;
  const bool mine_c = live && kq < m && x < n;
  if (mine_c && d.cin_off >= 0) cin = c_in[d.cin_off + kq + m * x];
  double acc = 0.0;
  constexpr int U = 8;
  for (int c0 = 0; __any(c0 < cnt); c0 += 16) {
    Entry own = Entry::make(0, 0, 0);  // k extent 0: a place past the end of the list feeds zeros
    if (c0 + t < cnt) own = e[c0 + t];
    const int left = cnt - c0;
    for (int q0 = 0; q0 < 16 && __any(q0 < left); q0 += U) {
      if constexpr (K4) {
        double av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int q = q0 + u;
          const int src = (sub * 4 + (q & 3) + 16 * (q >> 2)) * 4;
          Entry en;
          en.a_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)own.a_lo);
          en.b_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)own.b_lo);
          en.w = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)own.w);
          const int ks = en.ks();
          const bool kv = kq < ks;
          av[u] = (kv && x < m) ? a_data[en.a_off() + x + m * kq] : 0.0;
          bv[u] = (kv && x < n) ? b_data[en.b_off() + kq + ks * x] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av[u], bv[u], acc, 0, 0, 0);
      } else {
#pragma unroll 1
        for (int u = 0; u < U; ++u) {
          const int q = q0 + u;
          const int src = (sub * 4 + (q & 3) + 16 * (q >> 2)) * 4;
          Entry en;
          en.a_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)own.a_lo);
          en.b_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)own.b_lo);
          en.w = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)own.w);
          const int ks = en.ks();
          const double* A = a_data + en.a_off();
          const double* B = b_data + en.b_off();
          for (int kb = 0; __any(kb < ks); kb += 4) {
            const int k = kb + kq;
            const bool kv = k < ks;
            const double a = (kv && x < m) ? A[x + m * k] : 0.0;
            const double b = (kv && x < n) ? B[k + ks * x] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
          }
        }
      }
    }
  }
  if (!live || (skip_empty && cnt == 0)) return;
  if (mine_c) c_out[d.c_off + kq + m * x] = alpha * acc + (d.cin_off >= 0 ? beta * cin : 0.0);
}

}  // namespace dbcsr_amd
#endif
