// mm_band.hip -- numeric kernels of the band dataflow (mm_band.h): CU-wide C tiles, the B operand shared in an LDS ring.
// A translation unit of its own because it is compiled with -mllvm -structurizecfg-skip-uniform-regions (the product loop chooses one
// of nine accumulator sets with a wave-uniform switch; see mm_tile.hip and tests/test_kernel_resources.py).
#include "common.h"
#include "mm_types.h"
#include "smm_core.h"
#include "mm_band.h"

namespace dbcsr_amd {

__device__ __forceinline__ int64_t band_uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

constexpr unsigned kBandDone = 0x7fffffffu;  // position of a wave that needs nothing any more

// minimum over the wavefront as a scalar: DPP row shifts and row broadcasts (no LDS, no index registers), result from lane 63
__device__ __forceinline__ unsigned band_wave_min(unsigned v) {
  // (the control word must be a literal: one call per step)
  {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v = o < v ? o : v;
  }
  {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v = o < v ? o : v;
  }
  {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v = o < v ? o : v;
  }
  {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v = o < v ? o : v;
  }
  {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v = o < v ? o : v;
  }
  {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    v = o < v ? o : v;
  }
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int M, int N, int K, int D, int WAVES, int ASL>
struct BandKernel {
  static constexpr int KS = (K + 3) / 4;
  static constexpr int ABYTES = M * K * 8, BBYTES = K * N * 8;
  static constexpr int SA = (ABYTES + 15) & ~15, SB = (BBYTES + 15) & ~15;
  static constexpr int PA = (ABYTES + 1023) / 1024, PB = (BBYTES + 1023) / 1024;
  static constexpr int STATE = 512;  // D state words (D <= 64), then at + 256 the 64 words lanes 1..63 of an LDS atomic are sent to
  // [state][B ring: D slots][A rings: WAVES waves x ASL slots][slack: fragment reads of lanes past the last column / row of the last slot and
  // the whole 1 KiB pieces of the C staging read past a slot's end]
  static constexpr int LDS = STATE + D * SB + WAVES * ASL * SA + 1024;
  static_assert((M + 7) / 8 == 3 && (N + 7) / 8 == 3, "the sub-tiles are sized for blocks of 17..24 (9 accumulators per block and lane)");
  static_assert(D >= 4 && D <= 64, "ring depth");
  static_assert(PA == PB, "one wait count per copy");
  static_assert(PA + 2 * PB < 32, "vmcnt budget");
  static_assert(KS >= 3, "first / middle / last k step");
  static_assert(M * N * 8 <= SA, "a C block is staged in an A slot");
};

// state word of a ring slot: bits 6.. = sequence number of the block + D (so that "nothing yet" is a valid predecessor), bit 5 = the
// block has landed, bits 0-4 = waves that have not finished with it
__device__ __forceinline__ unsigned band_enc(unsigned seq_plus_d, unsigned landed, unsigned left) { return (seq_plus_d << 6) | (landed << 5) | left; }
constexpr unsigned kBandLanded = 32u;

// fragments of k step s: A from the wave's own slot, B from the shared ring.  One address register per operand, every fragment at a
// compile-time offset; single ds_read_b64 (volatile: never paired into ds_read2_b64, which costs 8 LDS cycles against 2 x 2).  No
// clamping: a lane whose row (column) is past the block reads a neighbouring element or whatever follows the slot and only pollutes
// accumulator rows (columns) that are never stored; in the last k step of a K that is not a multiple of 4 the lanes past the end get an
// exact zero on the A side and a finite B value (element (0, col + 1), or the zero padding the masked last DMA piece leaves).
template <int M, int N, int K>
__device__ __forceinline__ void band_frags(int s, const double* pa, const double* pb, bool ktail_dead, double (&av)[3], double (&bv)[3]) {
  constexpr int KS = (K + 3) / 4;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    av[a] = *(const volatile double __attribute__((address_space(3)))*)(pa + 8 * a + s * 4 * M);
    if (s == KS - 1 && (K & 3)) av[a] = ktail_dead ? 0.0 : av[a];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) bv[c] = *(const volatile double __attribute__((address_space(3)))*)(pb + 8 * K * c + 4 * s);
}

__device__ __forceinline__ void band_mfma9(double (&acc)[3][3], const double (&av)[3], const double (&bv)[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
}

// acc += A (own slot) x B (ring slot): two fragment stages, the reads of step s + 1 issued before the MFMAs of step s
template <int M, int N, int K>
__device__ __forceinline__ void band_multiply(double (&acc)[3][3], const double* pa, const double* pb, bool ktail_dead) {
  constexpr int KS = (K + 3) / 4;
  double av[2][3], bv[2][3];
  band_frags<M, N, K>(0, pa, pb, ktail_dead, av[0], bv[0]);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < KS) band_frags<M, N, K>(ks + 1, pa, pb, ktail_dead, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    band_mfma9(acc, av[ks & 1], bv[ks & 1]);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// C_out block <- alpha * acc (+ beta * C_in block), through an LDS area of the wave, in whole 1 KiB pieces with the streaming hint
template <int M, int N>
__device__ __forceinline__ void band_store_block(const double (&acc)[3][3], char* stage, int64_t c_off, int64_t cin_off, double* __restrict__ c_out,
                                                 const double* __restrict__ c_in, double alpha, double beta, const LaneMap& L, int voff) {
  constexpr int CC = (M * N * 8 + 1023) / 1024;
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  double* lds_c = reinterpret_cast<double*>(stage);
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
      if (row < M && col < N) lds_c[row + M * col] = alpha * acc[a][c];
    }
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(c_out + c_off), 0, M * N * 8, 0x00020000);
  if (cin_off >= 0) {
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(c_in + cin_off), 0, M * N * 8, 0x00020000);
    u32x4 ci[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) ci[c] = __builtin_amdgcn_raw_buffer_load_b128(rsi, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      f64x2 v = *reinterpret_cast<const f64x2*>(stage + c * 1024 + voff);
      const f64x2 w = __builtin_bit_cast(f64x2, ci[c]);
      v[0] += beta * w[0];
      v[1] += beta * w[1];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsc, voff + c * 1024, 0, 2);
    }
  } else {
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(stage + c * 1024 + voff);
      __builtin_amdgcn_raw_buffer_store_b128(v, rsc, voff + c * 1024, 0, 2);
    }
  }
}

// One persistent workgroup of 8 waves per CU; workgroup b works for XCD b % 8 (round-robin dispatch: a different placement costs L2
// hits, never correctness -- nothing here waits for another workgroup).  D: slots of the shared B ring; BPOL: cache policy of the B
// copies (dma_lds.h).
template <int M, int N, int K, int D, int BPOL, int WAVES, int TR, int TC, int ASL>
__global__ void __launch_bounds__(64 * WAVES) mm_numeric_f64_band(BandArgs P) {
  typedef BandKernel<M, N, K, D, WAVES, ASL> BK;
  constexpr int SLOTS = TR * TC;
  static_assert(ASL == 1 || ASL == 2, "A slots per wave");
  static_assert(SLOTS <= kBandSlots && WAVES <= kBandMaxWaves && WAVES * TR <= 32, "geometry the index kernels can express");
  constexpr int LOOK = 2;  // a wave tries to bring in the B block of the product after the next one: claimed at boundary p (after the A
                           // block of product p + 1 has been requested), published at boundary p + 2, when the wave's in-order wait for
                           // the A block of product p + 2 has covered the copy, used right then
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, voff = lane * 16;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
  const BandGeom G = P.G;
  if (cu >= G.cu_per_xcd) return;
  const unsigned state_lds = lds_offset_of(smem);
  const unsigned ringb_lds = state_lds + BK::STATE;
  const unsigned ringa_lds = ringb_lds + D * BK::SB + (unsigned)wid * (unsigned)ASL * BK::SA;
  char* const ringb = smem + BK::STATE;
  char* const ringa = ringb + D * BK::SB + wid * ASL * BK::SA;
  if (threadIdx.x < D) reinterpret_cast<unsigned*>(smem)[threadIdx.x] = band_enc(threadIdx.x, 1u, 0u);  // slot s: "block s - D" landed, nobody left
  __syncthreads();
  const LaneMap L(lane);
  const int la = L.rowl + M * L.kq, lb = L.kq + K * L.coll;  // lane parts of the fragment addresses (doubles)
  const bool ktail_dead = (K & 3) != 0 && (4 * (BK::KS - 1) + L.kq) >= K;
  // LDS atomics on the state words: one lane's worth of work, issued by the whole wave -- lanes 1..63 aim at a dump area of their own
  // (no divergent branch in the product loop: the compiler would restructure it around the choice of the accumulator set)
  const unsigned dump_lane = state_lds + 256u + 4u * (unsigned)lane;
  const bool lane0 = lane == 0;
  auto st_read = [&](unsigned s) -> unsigned {
    unsigned v;
    const unsigned addr = state_lds + 4u * s;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  };
  auto st_cas = [&](unsigned s, unsigned cmp, unsigned val) -> unsigned {
    unsigned old;
    const unsigned addr = lane0 ? state_lds + 4u * s : dump_lane;
    asm volatile("ds_cmpst_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(cmp), "v"(val) : "memory");
    return (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  };
  // (the LDS queue of a wave is in order: a decrement issued after the fragment reads of a block is carried out after them)
  auto st_add = [&](unsigned s, unsigned val) {
    const unsigned addr = lane0 ? state_lds + 4u * s : dump_lane;
    asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(val) : "memory");
  };
  auto st_or = [&](unsigned s, unsigned val) {
    const unsigned addr = lane0 ? state_lds + 4u * s : dump_lane;
    asm volatile("ds_or_b32 %0, %1" ::"v"(addr), "v"(val) : "memory");
  };
  const int g = xcd * G.cu_per_xcd + cu;
  const int64_t l0 = band_uniform64(P.list_off[(int64_t)(g * WAVES + wid) * G.max_i]);
  const int64_t l1 = band_uniform64(P.list_off[(int64_t)(g * WAVES + wid + 1) * G.max_i]);
  const int n = (int)(l1 - l0);
  const BandEntry* e = P.entries + l0;
  const __amdgpu_buffer_rsrc_t rs_list = __builtin_amdgcn_make_buffer_rsrc((void*)e, 0, n * 16, 0x00020000);
  const int64_t tile0 = G.lo(xcd) + cu;
  const bool timing = P.knobs & 1;
  unsigned long long t_a = 0, t_b = 0, t_mul = 0, t_epi = 0;
  unsigned n_late = 0, n_bwait = 0;
  const unsigned long long t_begin = timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
  double acc[SLOTS][3][3];
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[sl][a][c] = 0.0;
  struct Ent {
    uint32_t a_lo, b_lo, w, s;
    int t;  // tile of the workgroup's sweep the entry belongs to
  };
  const Ent nop = {0u, 0u, kBandNop, 0u, 0};
  Ent cur = nop, nxt = nop;  // the sweep starts LOOK entries early, on two non-products
  int aslot = 0;             // A slot of the current product
  unsigned pend1 = 0, pend2 = 0;  // slot + 1 of the B copies this wave issued at the previous boundary / the one before and has not published
  int nb_prev = 0;           // pieces of the B copy issued at the previous boundary
  int itile = 0;             // tiles of this workgroup the wave has written
  int tile_far = 0;          // tile of the entry that enters the look-ahead
  bool dead = false;         // a wait gave up (a bug): finish without waiting, the host reports it
  auto issue_b = [&](const Ent& en, unsigned slot) {
    const uint64_t bo = (uint64_t)en.b_lo | ((uint64_t)(en.w >> 24) << 32);
    dma_block<BK::BBYTES, BPOL>(P.b_data + bo, ringb_lds + slot * BK::SB, voff);
  };
  // (call only when the copies are known to have landed)
  auto publish_all = [&]() {
    if (pend2) st_or(pend2 - 1u, kBandLanded);
    if (pend1) st_or(pend1 - 1u, kBandLanded);
    pend1 = pend2 = 0;
  };
  // ---- the XCD's k window (as in mm_tile.hip): what a wave publishes is the sweep position of the next operands it will FETCH, a lower
  // bound of everything it still needs from L2; the wave that holds the minimum may always go on
  int window = __builtin_amdgcn_readfirstlane(P.window);
  constexpr int TEAM = 32 * WAVES;  // counters of an XCD's team: 4 (8) words per lane
  unsigned* team = P.prog + xcd * 512;
  const int q_team = cu * WAVES + wid;
  const int team_n = G.cu_per_xcd * WAVES < TEAM ? G.cu_per_xcd * WAVES : TEAM;  // (a multiple of 4)
  const __amdgpu_buffer_rsrc_t rs_team = __builtin_amdgcn_make_buffer_rsrc((void*)team, 0, 4 * TEAM, 0x00020000);
  const int pub_off = lane == 0 ? 4 * q_team : 0x7ffffff0;  // branch-free: the other lanes' stores are dropped by the bounds check
  const int qshift = (P.knobs >> 1) & 7 ? (P.knobs >> 1) & 7 : 3;
  const unsigned quantum = (window >> qshift) > 0 ? (unsigned)(window >> qshift) : 1u;
  unsigned seen_min = 0, published = 0;
  unsigned n_polls = 0, n_blocked = 0;
  unsigned long long t_win = 0;
  auto pos_of = [&](const Ent& en) -> unsigned { return (unsigned)en.t * (unsigned)G.kspan + (((en.w >> 8) & 255u) | ((en.s >> 28) << 8)); };
  auto publish_pos = [&](unsigned gp) {
    published = gp;
    __builtin_amdgcn_raw_buffer_store_b32(gp, rs_team, pub_off, 0, 0);  // into this XCD's L2 (where the whole team reads it)
  };
  auto team_min = [&]() -> unsigned {
    unsigned m = kBandDone;
#pragma unroll
    for (int h = 0; h < TEAM / 256; ++h) {
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_team, voff, 1024 * h, 16);  // sc1: not from this CU's vector cache
      if (256 * h + 4 * lane >= team_n) v = u32x4{kBandDone, kBandDone, kBandDone, kBandDone};
      unsigned m1 = v[0] < v[1] ? v[0] : v[1];
      const unsigned m2 = v[2] < v[3] ? v[2] : v[3];
      m1 = m1 < m2 ? m1 : m2;
      m = m < m1 ? m : m1;
    }
    return band_wave_min(m);
  };
  auto admit = [&](unsigned gp) {
    if (window <= 0) return;
    if (gp <= seen_min + (unsigned)window) {
      if (gp >= published + quantum) publish_pos(gp);
      return;
    }
    publish_pos(gp);
    ++n_blocked;
    const unsigned long long tw0 = timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (pend1 | pend2) {  // a waiting wave owes nothing
      dma_wait<0>();
      publish_all();
    }
    int polls = 0;
    for (;;) {
      ++n_polls;
      seen_min = team_min();
      if (gp <= seen_min + (unsigned)window) break;
      if (++polls > (1 << 15)) {  // never hang on the protocol: go on unthrottled (speed only)
        window = 0;
        atomicAdd(P.flags + 3, lane == 0 ? 1 : 0);
        publish_pos(kBandDone);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (timing) t_win += __builtin_amdgcn_s_memrealtime() - tw0;
  };
  // descriptors of the tile being swept, one vector load per tile, requested a whole tile ahead (right after the previous tile was
  // written): lane l < 9 holds c_off[l], lane 9 + l cin_off[l].  (Per-block scalar-looking loads inside the epilogue became vector loads
  // with a full s_waitcnt vmcnt(0) each -- the kernel stores to memory, so the compiler keeps them off the scalar cache -- and every C
  // block waited for the stores of the one before.)
  int64_t dval = -1;
  const int64_t tile_end = G.lo(xcd + 1);
  auto load_descs = [&]() {
    const int64_t T = tile0 + (int64_t)G.cu_per_xcd * itile;
    if (T >= tile_end) return;
    const int band = (int)(T / G.nBC), ct = (int)(T % G.nBC);
    const int64_t* tdp = reinterpret_cast<const int64_t*>(P.descs + ((int64_t)(WAVES * band + wid) * G.nBC + ct));
    dval = tdp[lane < 2 * kBandSlots ? lane : 2 * kBandSlots - 1];
  };
  load_descs();
  unsigned ev0 = 0u, ev1 = 0u, ev2 = 0u, ev3 = 0u;
  for (int wb = 0; wb < n + LOOK; wb += 64) {
    if (wb < n) {
      // entries wb .. wb + 63 in one vector load, handed out with v_readlane (no scalar load next to the fragment reads: SMEM returns out
      // of order and turns every lgkmcnt wait of the multiply into lgkmcnt(0)).  The wait is stated here, once per window: the compiler
      // must not place it inside the product loop, where it would also drain the copies in flight before every product.
      const u32x4 ev = __builtin_amdgcn_raw_buffer_load_b128(rs_list, voff, wb * 16, 0);
      ev0 = ev[0], ev1 = ev[1], ev2 = ev[2], ev3 = ev[3];
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(ev0), "+v"(ev1), "+v"(ev2), "+v"(ev3)::"memory");
    }
    const int jn = n + LOOK - wb < 64 ? n + LOOK - wb : 64;
    for (int j = 0; j < jn; ++j) {
      Ent far = nop;
      if (wb + j < n) {
        far.a_lo = (uint32_t)__builtin_amdgcn_readlane((int)ev0, j);
        far.b_lo = (uint32_t)__builtin_amdgcn_readlane((int)ev1, j);
        far.w = (uint32_t)__builtin_amdgcn_readlane((int)ev2, j);
        far.s = (uint32_t)__builtin_amdgcn_readlane((int)ev3, j);
        far.t = tile_far;
        if (far.w & kBandFlush) ++tile_far;
      }
      unsigned long long t0 = timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
      // (1) [two A slots] the A block of the next product, into the wave's other slot -- first the window: nobody fetches beyond the
      // team's minimum + W
      const bool next_a = !(nxt.w & kBandNop) && (nxt.w & kBandNewA);
      int nw = ASL == 2 ? nb_prev : 0;
      if (ASL == 2 && next_a) {
        admit(pos_of(nxt));
        if (timing) t0 = __builtin_amdgcn_s_memrealtime();
        const uint64_t ao = (uint64_t)nxt.a_lo | ((uint64_t)((nxt.w >> 16) & 0xffu) << 32);
        dma_block<BK::ABYTES>(P.a_data + ao, ringa_lds + (unsigned)(aslot ^ 1) * BK::SA, voff);
        nw += BK::PA;
      }
      // (2) the B block of the product after the next one: claim its slot if nobody has, the previous tenant is done with and the window
      // allows it (no waiting here: a wave that finds the block missing when it needs it fetches it then)
      unsigned newpend = 0;
      nb_prev = 0;
      if (!(far.w & kBandNop) && (window <= 0 || pos_of(far) <= seen_min + (unsigned)window)) {
        const unsigned seq = far.s & 0x7fffffu, slot = seq % (unsigned)D;
        const unsigned expect = band_enc(seq, 1u, 0u);
        if (st_cas(slot, expect, band_enc(seq + D, 0u, (far.s >> 23) & 31u)) == expect) {
          issue_b(far, slot);
          nw += BK::PB;
          nb_prev = BK::PB;
          newpend = slot + 1u;
        }
      }
      // (3) copies complete in order.  Two A slots: issued since the A block of the CURRENT product (first thing of the previous boundary)
      // are the B copy of the previous boundary and this boundary's A and B copies; when no more than those are in flight, the current A
      // block has landed, and so has the B copy issued two boundaries ago: it is published now -- two products of flight time, none of
      // it spent in this wait unless it was late.  One A slot: the current A block was requested at the end of the previous trip, after
      // that trip's B copy: only this trip's B copy is younger, and the previous trip's is published.
      if (nw == 0)
        dma_wait<0>();
      else if (nw == BK::PA)
        dma_wait<BK::PA>();
      else if (nw == 2 * BK::PA)
        dma_wait<2 * BK::PA>();
      else
        dma_wait<3 * BK::PA>();
      if (ASL == 2) {
        if (pend2) st_or(pend2 - 1u, kBandLanded);
        pend2 = pend1;
      } else {
        if (pend1) st_or(pend1 - 1u, kBandLanded);
      }
      pend1 = newpend;
      if (timing) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        t_a += t1 - t0;
        t0 = t1;
      }
      if (!(cur.w & kBandNop)) {
        // (4) the B block of the current product
        const unsigned seq = cur.s & 0x7fffffu, slot = seq % (unsigned)D;
        const unsigned want = (band_enc(seq + D, 1u, 0u)) >> 5;  // this block, landed (any number of users left)
        unsigned v = st_read(slot);
        if ((v >> 5) != want && !dead) {
          // not there yet.  A wave that waits publishes what it holds first (its own copy may be what it -- or the wave it waits for -- needs),
          // so a waiting wave never owes anything: whoever it waits for is running
          ++n_bwait;
          if (pend1 | pend2) {
            dma_wait<0>();
            publish_all();
            v = st_read(slot);
          }
          const unsigned expect = band_enc(seq, 1u, 0u);
          unsigned spins = 0;
          while ((v >> 5) != want) {
            if (v == expect) {  // free and unclaimed: bring it in now
              v = st_cas(slot, expect, band_enc(seq + D, 0u, (cur.s >> 23) & 31u));
              if (v == expect) {
                ++n_late;
                issue_b(cur, slot);
                dma_wait<0>();
                st_or(slot, kBandLanded);
                break;
              }
              continue;
            }
            if (++spins > (1u << 22)) {  // never hang: a protocol bug must end as a failed check, not as a dead GPU
              dead = true;
              atomicAdd(P.flags, lane == 0 ? 1 : 0);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
            v = st_read(slot);
          }
        }
        if (timing) {
          const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
          t_b += t1 - t0;
          t0 = t1;
        }
        // (5) multiply: one body per accumulator set (registers cannot be indexed)
        const double* pa = reinterpret_cast<const double*>(ringa + (ASL == 2 ? aslot : 0) * BK::SA) + la;
        const double* pb = reinterpret_cast<const double*>(ringb + slot * BK::SB) + lb;
        switch (cur.w & 15u) {
#define DBCSR_BAND_CASE(S_)                                                         \
  case S_:                                                                          \
    if constexpr (S_ < SLOTS) band_multiply<M, N, K>(acc[S_], pa, pb, ktail_dead);  \
    break;
          DBCSR_BAND_CASE(0) DBCSR_BAND_CASE(1) DBCSR_BAND_CASE(2) DBCSR_BAND_CASE(3) DBCSR_BAND_CASE(4) DBCSR_BAND_CASE(5)
          DBCSR_BAND_CASE(6) DBCSR_BAND_CASE(7) DBCSR_BAND_CASE(8)
#undef DBCSR_BAND_CASE
          default: break;
        }
        // (6) last product of this wave with the block: one user less
        if (cur.w & kBandLastB) st_add(slot, 0xffffffffu);
        if (timing) t_mul += __builtin_amdgcn_s_memrealtime() - t0;
      }
      if (cur.w & kBandFlush) {
        // end of a tile: write the sub-tile's C blocks (staged in the A slot of the last product: its fragments have been read) and
        // go on with the next tile of the workgroup -- the other waves may still be in this one
        const unsigned long long t_e0 = timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
        char* stage = ringa + (ASL == 2 ? aslot : 0) * BK::SA;
        const unsigned dlo = (unsigned)dval, dhi = (unsigned)((uint64_t)dval >> 32);
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
          const int64_t c_off = (int64_t)(((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)dhi, sl) << 32) | (unsigned)__builtin_amdgcn_readlane((int)dlo, sl));
          const int64_t cin_off = (int64_t)(((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)dhi, kBandSlots + sl) << 32) |
                                            (unsigned)__builtin_amdgcn_readlane((int)dlo, kBandSlots + sl));
          if (c_off >= 0) band_store_block<M, N>(acc[sl], stage, c_off, cin_off, P.c_out, P.c_in, P.alpha, P.beta, L, voff);
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[sl][a][c] = 0.0;
        }
        ++itile;
        load_descs();
        if (timing) t_epi += __builtin_amdgcn_s_memrealtime() - t_e0;
      }
      if (ASL == 2) {
        if (next_a) aslot ^= 1;
      } else if (next_a) {
        // (7) [one A slot] the fragments of the current product have been read (its MFMAs were issued), a tile's C blocks have left the
        // slot: request the next product's A block now -- the other waves of the SIMD multiply meanwhile
        admit(pos_of(nxt));
        const uint64_t ao = (uint64_t)nxt.a_lo | ((uint64_t)((nxt.w >> 16) & 0xffu) << 32);
        dma_block<BK::ABYTES>(P.a_data + ao, ringa_lds, voff);
      }
      cur = nxt;
      nxt = far;
    }
  }
  if (window > 0) publish_pos(kBandDone);
  dma_wait<0>();
  publish_all();
  if (timing && P.times) {
    const unsigned long long t_all = __builtin_amdgcn_s_memrealtime() - t_begin;
    if (lane == 0) {
      atomicAdd(P.times + 0, t_all);
      atomicAdd(P.times + 1, t_a);
      atomicAdd(P.times + 2, t_b);
      atomicAdd(P.times + 3, t_mul);
      atomicAdd(P.times + 4, t_epi);
      atomicAdd(P.times + 5, 1ull);
      atomicAdd(P.times + 6, (unsigned long long)n_late);
      atomicAdd(P.times + 7, (unsigned long long)n_bwait);
      atomicAdd(P.times + 8, t_win);
      atomicAdd(P.times + 9, (unsigned long long)n_blocked);
      atomicAdd(P.times + 10, (unsigned long long)n_polls);
    }
  }
}

// products of the tiles' C blocks whose inner block has another size than K (the tail block column of A): C += alpha * A * B on the
// finished block, one wavefront per sub-tile (all products of a C block are in one list: no two waves touch a block)
template <int M, int N>
__global__ void __launch_bounds__(256) band_remainder(int64_t nsub, const BandDesc* __restrict__ descs, const int64_t* __restrict__ rem_start,
                                                      const BandRem* __restrict__ rem, const double* __restrict__ a_data,
                                                      const double* __restrict__ b_data, double* __restrict__ c_out, double alpha) {
  const int lane = threadIdx.x & 63;
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (t >= nsub) return;
  const int64_t r0 = rem_start[t], r1 = rem_start[t + 1];
  if (r1 == r0) return;
  const BandDesc* td = descs + t;
  const LaneMap L(lane);
  for (int64_t p = r0; p < r1; ++p) {
    const BandRem en = rem[p];
    const uint64_t ao = (uint64_t)en.a_lo | ((uint64_t)((en.w >> 16) & 0xffu) << 32), bo = (uint64_t)en.b_lo | ((uint64_t)(en.w >> 24) << 32);
    const int ks = (int)((en.w >> 8) & 0xffu), slot = (int)(en.w & 15u);
    double acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[a][c] = 0.0;
    block_product_f64<3, 3, false>(acc, a_data + ao, b_data + bo, M, N, ks, L);
    double* C = c_out + td->c_off[slot];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
        if (row < M && col < N) C[row + (size_t)M * col] += alpha * acc[a][c];
      }
  }
}

bool band_shape(int shape, int* waves, int* tr, int* tc) {
  switch (shape) {
    case 0: *waves = 8, *tr = 3, *tc = 3; return true;
    case 1: *waves = 16, *tr = 2, *tc = 2; return true;
    default: return false;
  }
}
// (WAVES, TR, TC, ASL) of the shapes, as template arguments
#define DBCSR_BAND_SHAPE0 8, 3, 3, 2
#define DBCSR_BAND_SHAPE1 16, 2, 2, 1

// ring depths the kernel is built for (LDS: 512 + D x 4240 + 67840 + 1024 bytes for 23 x 23 blocks in both shapes; 160 KB per CU)
#define DBCSR_AMD_BAND_DEPTHS(X, S_) X(S_, 12) X(S_, 16) X(S_, 20) X(S_, 22)

int band_lds_bytes(int m, int n, int k, int shape, int depth) {
  if (m != n || m != k || (shape != 0 && shape != 1)) return 0;
  switch (m) {
#define DBCSR_BAND_LDS_D(S_, D_) \
  if (depth == D_) return shape == 1 ? BandKernel<S_, S_, S_, D_, 16, 1>::LDS : BandKernel<S_, S_, S_, D_, 8, 2>::LDS;
#define DBCSR_BAND_LDS(S_)                           \
  case S_:                                           \
    DBCSR_AMD_BAND_DEPTHS(DBCSR_BAND_LDS_D, S_)      \
    return 0;
    DBCSR_AMD_BAND_SIZES(DBCSR_BAND_LDS)
#undef DBCSR_BAND_LDS
#undef DBCSR_BAND_LDS_D
    default: return 0;
  }
}

template <int S_, int D_, int BPOL, int WAVES, int TR, int TC, int ASL>
static int band_launch_one(unsigned nwg, hipStream_t st, const BandArgs& P) {
  typedef BandKernel<S_, S_, S_, D_, WAVES, ASL> BK;
  static bool attr = false;
  if (!attr) {  // more than 64 KB of dynamic LDS needs the attribute, once per kernel
    ACC_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mm_numeric_f64_band<S_, S_, S_, D_, BPOL, WAVES, TR, TC, ASL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BK::LDS));
    attr = true;
  }
  hipLaunchKernelGGL((mm_numeric_f64_band<S_, S_, S_, D_, BPOL, WAVES, TR, TC, ASL>), dim3(nwg), dim3(64 * WAVES), BK::LDS, st, P);
  return check(hipGetLastError(), "mm_numeric_f64_band", __FILE__, __LINE__);
}

int band_launch(int m, int n, int k, int shape, int depth, int bpol, unsigned nwg, hipStream_t st, const BandArgs& P) {
  if (m != n || m != k) return 1;
  switch (m) {
#define DBCSR_BAND_LAUNCH_D(S_, D_)                                                                                                         \
  if (depth == D_) {                                                                                                                        \
    if (shape == 1) return bpol == 1 ? band_launch_one<S_, D_, 1, DBCSR_BAND_SHAPE1>(nwg, st, P) : band_launch_one<S_, D_, 0, DBCSR_BAND_SHAPE1>(nwg, st, P); \
    return bpol == 1 ? band_launch_one<S_, D_, 1, DBCSR_BAND_SHAPE0>(nwg, st, P) : band_launch_one<S_, D_, 0, DBCSR_BAND_SHAPE0>(nwg, st, P);           \
  }
#define DBCSR_BAND_LAUNCH(S_)                        \
  case S_:                                           \
    DBCSR_AMD_BAND_DEPTHS(DBCSR_BAND_LAUNCH_D, S_)   \
    return 1;
    DBCSR_AMD_BAND_SIZES(DBCSR_BAND_LAUNCH)
#undef DBCSR_BAND_LAUNCH
#undef DBCSR_BAND_LAUNCH_D
    default: return 1;
  }
}

int band_launch_remainder(int m, int n, hipStream_t st, int64_t nsub, const BandDesc* descs, const int64_t* rem_start, const BandRem* rem,
                          const double* a_data, const double* b_data, double* c_out, double alpha) {
  if (m != n) return 1;
  const dim3 grid((unsigned)((nsub * 64 + 255) / 256));
  switch (m) {
#define DBCSR_BAND_REM(S_)                                                                                                          \
  case S_:                                                                                                                          \
    hipLaunchKernelGGL((band_remainder<S_, S_>), grid, dim3(256), 0, st, nsub, descs, rem_start, rem, a_data, b_data, c_out, alpha); \
    return check(hipGetLastError(), "band_remainder", __FILE__, __LINE__);
    DBCSR_AMD_BAND_SIZES(DBCSR_BAND_REM)
#undef DBCSR_BAND_REM
    default: return 1;
  }
}

}  // namespace dbcsr_amd
