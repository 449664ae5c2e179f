// mm_tile.hip -- the numeric kernels of the tile dataflow (mm_tile.h).  A translation unit of its own because it is compiled with
// -mllvm -structurizecfg-skip-uniform-regions: the product loop chooses one of nine accumulator sets with a wave-uniform switch,
// and without the option the compiler restructures that switch into a chain of flow blocks whose joins copy whole accumulator
// sets (and spill some): 400 v_mov_b64 and 116 bytes of scratch against none (tests/test_kernel_resources.py pins it).
#include "common.h"
#include "mm_types.h"
#include "smm_core.h"
#include "mm_tile.h"

namespace dbcsr_amd {

// ---- numeric kernel --------------------------------------------------------------------------------------------------


// minimum over the wavefront as a scalar: DPP row shifts and row broadcasts (no LDS, no index registers), result from lane 63
__device__ __forceinline__ unsigned tile_wave_min(unsigned v) {
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

__device__ __forceinline__ int64_t tile_uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int M, int N, int K, int RDV, int DEPTH = 2>
struct TileKernel {
  static constexpr int MA = (M + 7) / 8, NC = (N + 7) / 8, KS = (K + 3) / 4;
  static constexpr int ABYTES = M * K * 8, BBYTES = K * N * 8;
  static constexpr int SA = (ABYTES + 15) & ~15, SB = (BBYTES + 15) & ~15, SLOT = SA + SB;
  static constexpr int PIECES = (ABYTES + 1023) / 1024 + (BBYTES + 1023) / 1024;
  static constexpr int CBYTES = ((M * N * 8 + 1023) / 1024) * 1024;
  // per wave: DEPTH slots; the C blocks leave through the ring itself when it has two slots, through a staging area of their own
  // behind a deeper one.  + 64: fragment reads of lanes past the last column, whole 16-byte lanes of the C staging
  static constexpr int STAGE = DEPTH > 2 ? DEPTH * SLOT : 0;
  static constexpr int RING = (DEPTH > 2 ? DEPTH * SLOT + CBYTES : (2 * SLOT > CBYTES ? 2 * SLOT : CBYTES)) + 64;
  static_assert(MA == 3 && NC == 3, "the sub-tiles are sized for blocks of 17..24 (9 accumulators per block and lane)");
  static_assert((DEPTH - 1) * PIECES < 64, "vmcnt budget");
  static_assert((DEPTH & (DEPTH - 1)) == 0, "ring slots are addressed with a mask");
  static_assert(KS >= 3, "first / middle / last k step");
};

// Fragments of k step s of the product staged in a ring slot.  ONE address register per operand: pa = slot + lane part of A,
// pb = slot + lane part of B, every fragment at a compile-time offset from them.  No clamping: a lane whose row (column) is past
// the block reads the neighbouring element -- finite data of the block, or the slot's padding, which the masked last DMA piece
// fills with the zeros of its out-of-range bytes -- and only pollutes accumulator rows (columns) that are never stored.  In the
// last k step of a K that is not a multiple of 4 the lanes past the end get an exact zero on the A side and a finite B value
// (element (0, col + 1), or the zero padding after the last column).
template <int M, int N, int K, int RDV>
__device__ __forceinline__ void tile_frags(int s, const double* pa, const double* pb, bool ktail_dead, double (&av)[3], double (&bv)[3]) {
  typedef TileKernel<M, N, K, RDV> TK;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if constexpr (RDV == 1)
      av[a] = *(const volatile double __attribute__((address_space(3)))*)(pa + 8 * a + s * 4 * M);
    else
      av[a] = pa[8 * a + s * 4 * M];
    if (s == TK::KS - 1 && (K & 3)) av[a] = ktail_dead ? 0.0 : av[a];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if constexpr (RDV == 1)
      bv[c] = *(const volatile double __attribute__((address_space(3)))*)(pb + 8 * K * c + 4 * s);
    else
      bv[c] = pb[8 * K * c + 4 * s];
  }
}

// nine MFMAs of one k step, accumulators updated IN PLACE.  Written as inline asm with tied ("+v") accumulator operands: with the
// builtin, each MFMA defines a new value, and across the nine-way choice of the accumulator set the compiler kept two register
// homes per set and copied whole sets at every join (dozens of v_mov_b64 per product).  Dependent MFMAs on the same accumulator
// are nine instructions apart, as in the compiler's own sequences (no software wait states needed at that distance); the s_nop
// covers a fragment register written by a VALU select just before (k tail).
__device__ __forceinline__ void tile_mfma9(double (&acc)[3][3], const double (&av)[3], const double (&bv)[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
}

// acc += the product staged in a ring slot.  Two fragment stages: the reads of step s + 1 are issued, then the MFMAs of step s run
// (the asm statements are volatile: they keep their order, and the fragment reads cannot sink below the statement that uses them)
template <int M, int N, int K, int RDV>
__device__ __forceinline__ void tile_multiply(double (&acc)[3][3], const double* pa, const double* pb, bool ktail_dead) {
  typedef TileKernel<M, N, K, RDV> TK;
  double av[2][3], bv[2][3];
  tile_frags<M, N, K, RDV>(0, pa, pb, ktail_dead, av[0], bv[0]);
#pragma unroll
  for (int ks = 0; ks < TK::KS; ++ks) {
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < TK::KS) tile_frags<M, N, K, RDV>(ks + 1, pa, pb, ktail_dead, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    tile_mfma9(acc, av[ks & 1], bv[ks & 1]);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// C_out block <- alpha * acc (+ beta * C_in block), through the wave's LDS area, in whole 1 KiB pieces with the streaming hint
template <int M, int N>
__device__ __forceinline__ void tile_store_block(const double (&acc)[3][3], char* stage, int64_t c_off, int64_t cin_off, double* __restrict__ c_out,
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

// One persistent workgroup per CU, WAVES waves; workgroup b belongs to the team of XCD b % 8 (round-robin dispatch; a different
// placement costs L2 hits, never correctness).  TR x TC: C blocks of a wave's sub-tile; DEPTH: ring slots per wave (the product being
// multiplied + DEPTH - 1 in flight).
template <int M, int N, int K, int RDV, int TR, int TC, int WAVES, int DEPTH>
__global__ void __launch_bounds__(64 * WAVES) mm_numeric_f64_tile(TileArgs P) {
  typedef TileKernel<M, N, K, RDV, DEPTH> TK;
  constexpr int SLOTS = TR * TC, AHEAD = DEPTH - 1;
  static_assert(SLOTS <= kTileMaxSlots && TR <= kTileMaxT && TC <= kTileMaxT, "slot numbering of the index kernels");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, voff = lane * 16;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
  const int q = cu * WAVES + wid;  // position in the team
  char* ring = smem + wid * TK::RING;
  char* stage = ring + TK::STAGE;
  const unsigned ring_lds = lds_offset_of(ring);
  const LaneMap L(lane);
  const TileGeom G = P.G;
  unsigned* team = P.prog + xcd * 256;
  // lane parts of the fragment addresses inside a ring slot (doubles): constant for the whole life of the wave
  const int la = L.rowl + M * L.kq, lb = TK::SA / 8 + L.kq + K * L.coll;
  const bool ktail_dead = (K & 3) != 0 && (4 * (TK::KS - 1) + L.kq) >= K;
  const int team_n = G.team_rows * kTeamCols < 256 ? G.team_rows * kTeamCols : 256;
  const bool in_team = q < team_n;
  const bool my_counters = 4 * lane < team_n;  // (team_n is a multiple of 4: kTeamCols is)
  int window = __builtin_amdgcn_readfirstlane(P.window);
  unsigned seen_min = 0;  // last minimum read from the team (the true minimum only grows: a stale value is conservative)
  const __amdgpu_buffer_rsrc_t rs_team = __builtin_amdgcn_make_buffer_rsrc((void*)team, 0, 1024, 0x00020000);
  // Publishing is branch-free: every lane issues the store, the lanes other than 0 with an offset past the end of the buffer
  // descriptor (dropped by the bounds check).  A divergent branch inside the product loop would make the compiler restructure the
  // whole loop body -- including the choice of the accumulator set, whose joins then copy whole sets.
  const int pub_off = lane == 0 ? 4 * q : 0x7ffffff0;
  // Stores to the team's counters are rationed: the team's waves share eight cache lines, and a store per product (29 M per
  // multiply of config 2, every one a partial-line write) took longer than the multiply itself.  A wave republishes only when its
  // need has moved on by a quantum (an eighth of the window): what it publishes is a LOWER bound of what it still needs, so
  // publishing late is always safe, and the team's view of it lags by less than the quantum.
  const int qshift = P.knobs & 7 ? (P.knobs & 7) : 3;             // knobs bits 0-2: publish quantum = window >> qshift (default 3)
  const bool use_prio = (P.knobs >> 3) & 1;                       // bit 3: issue priority by distance from the team's minimum
  const unsigned quantum = (window >> qshift) > 0 ? (unsigned)(window >> qshift) : 1u;
  unsigned published = 0;
  unsigned pf_sink = 0;             // destination of the L2 prefetch loads (see the product loop)
  const int pf_off = lane * 128;    // one dword per 128-byte line
  unsigned n_polls = 0, n_blocked = 0;  // diagnostics: reads of the team's counters / products that had to wait for the window
  // knob bit 5: where a wave's time goes (s_memrealtime ticks of 10 ns): window waits, waits for operands, multiplies, epilogues
  const bool timing = (P.knobs >> 5) & 1;
  unsigned long long t_admit = 0, t_wait = 0, t_mul = 0, t_epi = 0, t_all = 0;
  const unsigned long long t_begin = timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
  auto publish = [&](unsigned g) {
    published = g;
    if (P.pub_policy == 0)
      __builtin_amdgcn_raw_buffer_store_b32(g, rs_team, pub_off, 0, 16);  // sc1: written through, visible device-wide
    else
      __builtin_amdgcn_raw_buffer_store_b32(g, rs_team, pub_off, 0, 0);   // into this XCD's L2 (where the whole team reads it)
  };
  // next-need protocol: the wave that holds the minimum may always go on.  What a wave publishes is the k of the next product it
  // will FETCH (what is already on its way to LDS needs no L2 residency any more).
  auto admit = [&](unsigned g) {
    if (window <= 0) return;
    if (g <= seen_min + (unsigned)window) {  // inside the window already: no traffic at all, except the rationed progress report
      if (g >= published + quantum) publish(g);
      if (use_prio) {
        const unsigned lead = g - seen_min;   // (seen_min is a stale lower bound of the minimum: lead is an upper bound)
        if (lead * 4 < (unsigned)window)
          __builtin_amdgcn_s_setprio(3);
        else if (lead * 4 > 3u * (unsigned)window)
          __builtin_amdgcn_s_setprio(0);
        else
          __builtin_amdgcn_s_setprio(1);
      }
      return;
    }
    publish(g);
    int polls = 0;
    ++n_blocked;
    for (;;) {
      ++n_polls;
      // the counters of the team in one 1 KiB read (sc1: not from this CU's vector cache); lanes past the team see "done"
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_team, voff, 0, 16);
      if (!my_counters) v = u32x4{kTileDone, kTileDone, kTileDone, kTileDone};
      unsigned m = v[0] < v[1] ? v[0] : v[1];
      const unsigned m2 = v[2] < v[3] ? v[2] : v[3];
      m = m < m2 ? m : m2;
      seen_min = tile_wave_min(m);
      if (g <= seen_min + (unsigned)window) break;
      if (++polls > (1 << 15)) {  // never hang on the protocol: go on unthrottled
        window = 0;
        atomicAdd(P.flags, lane == 0 ? 1 : 0);
        publish(kTileDone);
        break;
      }
      if ((P.knobs >> 4) & 1)
        __builtin_amdgcn_s_sleep(1);
      else
        __builtin_amdgcn_s_sleep(2);
    }
  };
  double acc[SLOTS][3][3];
  for (int s = 0; in_team && s < G.nseq; ++s) {
    const int st = xcd + 8 * s;
    if (st >= G.nSR * G.nSC) break;
    const int tr = (st / G.nSC) * G.team_rows + q / kTeamCols, tc = (st % G.nSC) * kTeamCols + q % kTeamCols;
    if (tr >= G.nTR || tc >= G.nTC) {  // no sub-tile here (edge of the matrix): do not hold the team back
      if (window > 0) publish((unsigned)((s + 1) * G.kspan));
      continue;
    }
    const TileDesc* td = P.tdescs + ((int64_t)tr * G.nTC + tc);
    const int n = __builtin_amdgcn_readfirstlane(td->n_main);
    const int64_t ls = tile_uniform64(td->list_start);
    const TileEntry* e = P.entries + ls;
    const unsigned kbase = (unsigned)(s * G.kspan);
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl)
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[sl][a][c] = 0.0;
    // The list is read in WINDOWS of 64 entries with one vector load (lane l holds entry base + l) and handed out with v_readlane:
    // no scalar load next to the fragment reads (SMEM returns out of order: one in flight turns every lgkmcnt wait of the
    // multiply into lgkmcnt(0) -- measured: 22.4 -> 27.2 ms), and no vector load INSIDE the loop over a window's products (the
    // compiler would have to place s_waitcnt vmcnt(0) where a refill branch joins, i.e. before every product, and that wait also
    // covers the LDS-DMA pieces in flight).  Windows overlap by AHEAD entries: product p requests the operands of product p + AHEAD.
    const __amdgpu_buffer_rsrc_t rs_list = __builtin_amdgcn_make_buffer_rsrc((void*)e, 0, n * 16, 0x00020000);
    u32x4 ev = {0u, 0u, 0u, 0u};
    auto entry_of = [&](int j) {  // entry base + j of the current window, j wave-uniform
      TileEntry en;
      en.a_lo = (uint32_t)__builtin_amdgcn_readlane((int)ev[0], j);
      en.b_lo = (uint32_t)__builtin_amdgcn_readlane((int)ev[1], j);
      en.w = (uint32_t)__builtin_amdgcn_readlane((int)ev[2], j);
      en.k = (uint32_t)__builtin_amdgcn_readlane((int)ev[3], j);
      return en;
    };
    auto issue = [&](const TileEntry& en, int slot) {
      const uint64_t ao = (uint64_t)en.a_lo | ((uint64_t)((en.w >> 16) & 0xffu) << 32), bo = (uint64_t)en.b_lo | ((uint64_t)(en.w >> 24) << 32);
      const unsigned lds = ring_lds + (unsigned)slot * TK::SLOT;
      dma_block<TK::ABYTES>(P.a_data + ao, lds, voff);
      dma_block<TK::BBYTES>(P.b_data + bo, lds + TK::SA, voff);
    };
    constexpr int STEP = 64 - AHEAD;
    for (int base = 0; base < n; base += STEP) {
      ev = __builtin_amdgcn_raw_buffer_load_b128(rs_list, voff, base * 16, 0);  // entries base .. base + 63 (zeros past the end)
      const int jn = n - base < STEP ? n - base : STEP;
      if (base == 0) {  // the first AHEAD products of the list
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
          if (i < n) {
            const TileEntry en = entry_of(i);
            admit(kbase + en.k);
            issue(en, i);
          }
      }
      for (int j = 0; j < jn; ++j) {
        const int p = base + j;
        unsigned long long t0 = timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const uint32_t cur_w = (uint32_t)__builtin_amdgcn_readlane((int)ev[2], j);
        if (p + AHEAD < n) {
          const TileEntry nxt = entry_of(j + AHEAD);
          admit(kbase + nxt.k);
          if (timing) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            t_admit += t1 - t0;
            t0 = t1;
          }
          issue(nxt, (p + AHEAD) & (DEPTH - 1));
          if (DEPTH == 2 && P.prefetch && j + 2 < 64 && p + 2 < n) {
            // pull the blocks of product p + 2 into this XCD's L2 ahead of their DMA (one dword per 128-byte line, lanes past the
            // block are dropped by the descriptor's bounds check; the loaded values are never used): the two-slot ring holds one
            // product in flight per wave, which covers an L2 hit but not a trip over the fabric
            const TileEntry pf = entry_of(j + 2);
            const uint64_t ao = (uint64_t)pf.a_lo | ((uint64_t)((pf.w >> 16) & 0xffu) << 32), bo = (uint64_t)pf.b_lo | ((uint64_t)(pf.w >> 24) << 32);
            const dma_rsrc_t ra = dma_make_rsrc(P.a_data + ao, (unsigned)TK::ABYTES), rb = dma_make_rsrc(P.b_data + bo, (unsigned)TK::BBYTES);
            // (inline asm: the compiler must never wait for these loads; they all land in ONE register that nothing else may use,
            // kept alive until the wave's last wait)
            asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen sc1\n\tbuffer_load_dword %0, %1, %3, 0 offen sc1"
                         : "+v"(pf_sink)
                         : "v"(pf_off), "s"(ra), "s"(rb)
                         : "memory");
            dma_wait<TK::PIECES + 2>();
          } else {
            dma_wait<AHEAD * TK::PIECES>();  // the pieces of product p have landed (loads return in order; a pending store only makes this wait longer)
          }
        } else {
          // the tail of the list: fewer products in flight behind this one
          const int newer = n - 1 - p;  // < AHEAD
          if (AHEAD >= 3 && newer == 2)
            dma_wait<2 * TK::PIECES>();
          else if (AHEAD >= 2 && newer == 1)
            dma_wait<1 * TK::PIECES>();
          else
            dma_wait<0>();
        }
        if (timing) {
          const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
          t_wait += t1 - t0;
          t0 = t1;
        }
        const double* sl = reinterpret_cast<const double*>(ring + (p & (DEPTH - 1)) * TK::SLOT);
        const double *pa = sl + la, *pb = sl + lb;
        // one multiply body per accumulator set (the set is a compile-time choice: registers cannot be indexed)
        switch (cur_w & 15u) {
#define DBCSR_TILE_CASE(S_)                                                                     \
  case S_:                                                                                      \
    if constexpr (S_ < SLOTS) tile_multiply<M, N, K, RDV>(acc[S_], pa, pb, ktail_dead);         \
    break;
          DBCSR_TILE_CASE(0) DBCSR_TILE_CASE(1) DBCSR_TILE_CASE(2) DBCSR_TILE_CASE(3) DBCSR_TILE_CASE(4) DBCSR_TILE_CASE(5)
          DBCSR_TILE_CASE(6) DBCSR_TILE_CASE(7) DBCSR_TILE_CASE(8) DBCSR_TILE_CASE(9) DBCSR_TILE_CASE(10) DBCSR_TILE_CASE(11)
          DBCSR_TILE_CASE(12) DBCSR_TILE_CASE(13) DBCSR_TILE_CASE(14)
#undef DBCSR_TILE_CASE
          default:
            if constexpr (15 < SLOTS) tile_multiply<M, N, K, RDV>(acc[SLOTS - 1], pa, pb, ktail_dead);
            break;
        }
        if (timing) t_mul += __builtin_amdgcn_s_memrealtime() - t0;
      }
    }
    const unsigned long long t_e0 = timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // the wave needs nothing below the next super-tile any more: do not hold the team back during the epilogue
    if (window > 0) publish((unsigned)((s + 1) * G.kspan));
#pragma unroll
    for (int sl2 = 0; sl2 < SLOTS; ++sl2) {
      // (wave-uniform by construction; said explicitly so that the buffer descriptors made from them stay in scalar registers)
      const int64_t c_off = tile_uniform64(td->c_off[sl2]), cin_off = tile_uniform64(td->cin_off[sl2]);
      if (c_off >= 0) tile_store_block<M, N>(acc[sl2], stage, c_off, cin_off, P.c_out, P.c_in, P.alpha, P.beta, L, voff);
    }
    if (timing) t_epi += __builtin_amdgcn_s_memrealtime() - t_e0;
  }
  if (q < 256) publish(kTileDone);
  dma_wait<0>();
  asm volatile("" ::"v"(pf_sink));
  if (timing && P.times) {
    t_all = __builtin_amdgcn_s_memrealtime() - t_begin;
    if (lane == 0) {
      atomicAdd(P.times + 0, t_all);
      atomicAdd(P.times + 1, t_admit);
      atomicAdd(P.times + 2, t_wait);
      atomicAdd(P.times + 3, t_mul);
      atomicAdd(P.times + 4, t_epi);
      atomicAdd(P.times + 5, 1ull);
    }
  }
  if (P.window > 0) {
    atomicAdd(P.flags + 2, lane == 0 ? (int)(n_polls >> 4) : 0);   // (in units of 16: the sum over the waves stays in range)
    atomicAdd(P.flags + 3, lane == 0 ? (int)(n_blocked >> 4) : 0);
  }
}

// products of the tiles' C blocks whose inner block has another size than K (the tail block column of A): C += alpha * A * B on
// the finished block, one wavefront per sub-tile (all products of a C block are in one list: no two waves touch a block)
template <int M, int N>
__global__ void __launch_bounds__(256) tile_remainder(TileGeom G, const TileDesc* __restrict__ tdescs, const TileEntry* __restrict__ entries,
                                                      const double* __restrict__ a_data, const double* __restrict__ b_data, double* __restrict__ c_out,
                                                      double alpha) {
  const int lane = threadIdx.x & 63;
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (t >= (int64_t)G.nTR * G.nTC) return;
  const TileDesc* td = tdescs + t;
  const int n_rem = td->n_rem;
  if (n_rem == 0) return;
  const LaneMap L(lane);
  const TileEntry* e = entries + td->list_start + td->n_main;
  for (int p = 0; p < n_rem; ++p) {
    const TileEntry en = e[p];
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


bool tile_shape(int shape, int* tr, int* tc, int* wg_waves) {
  switch (shape) {
    case 0: *tr = 3, *tc = 3, *wg_waves = 8; return true;
    case 1: *tr = 4, *tc = 3, *wg_waves = 4; return true;
    default: return false;
  }
}

// (TR, TC, WAVES, DEPTH) of the shapes: the table of tile_shape(), as template arguments
#define DBCSR_TILE_SHAPE0 3, 3, 8, 2
#define DBCSR_TILE_SHAPE1 4, 3, 4, 4

int tile_lds_bytes(int m, int n, int k, int shape) {
  if (m != n || m != k) return 0;
  switch (m) {
#define DBCSR_TILE_LDS(S_) \
  case S_: return shape == 1 ? 4 * TileKernel<S_, S_, S_, 0, 4>::RING : 8 * TileKernel<S_, S_, S_, 0, 2>::RING;
    DBCSR_AMD_TILE_SIZES(DBCSR_TILE_LDS)
#undef DBCSR_TILE_LDS
    default: return 0;
  }
}

template <int S_, int RDV, int TR, int TC, int WAVES, int DEPTH>
static int tile_launch_one(unsigned nwg, hipStream_t st, const TileArgs& P) {
  typedef TileKernel<S_, S_, S_, RDV, DEPTH> TK;
  static bool attr = false;
  if (!attr) {  // more than 64 KB of dynamic LDS needs the attribute, once per kernel
    ACC_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mm_numeric_f64_tile<S_, S_, S_, RDV, TR, TC, WAVES, DEPTH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, WAVES * TK::RING));
    attr = true;
  }
  hipLaunchKernelGGL((mm_numeric_f64_tile<S_, S_, S_, RDV, TR, TC, WAVES, DEPTH>), dim3(nwg), dim3(64 * WAVES), WAVES * TK::RING, st, P);
  return check(hipGetLastError(), "mm_numeric_f64_tile", __FILE__, __LINE__);
}

int tile_launch(int m, int n, int k, int rdv, int shape, unsigned nwg, hipStream_t st, const TileArgs& P) {
  if (m != n || m != k) return 1;
  switch (m) {
#define DBCSR_TILE_LAUNCH(S_)                                                                                                         \
  case S_:                                                                                                                            \
    if (shape == 1) return rdv ? tile_launch_one<S_, 1, DBCSR_TILE_SHAPE1>(nwg, st, P) : tile_launch_one<S_, 0, DBCSR_TILE_SHAPE1>(nwg, st, P); \
    return rdv ? tile_launch_one<S_, 1, DBCSR_TILE_SHAPE0>(nwg, st, P) : tile_launch_one<S_, 0, DBCSR_TILE_SHAPE0>(nwg, st, P);
    DBCSR_AMD_TILE_SIZES(DBCSR_TILE_LAUNCH)
#undef DBCSR_TILE_LAUNCH
    default: return 1;
  }
}

int tile_launch_remainder(int m, int n, hipStream_t st, const TileGeom& G, const TileDesc* tdescs, const TileEntry* entries, const double* a_data,
                          const double* b_data, double* c_out, double alpha) {
  if (m != n) return 1;
  const int64_t nT = (int64_t)G.nTR * G.nTC;
  const dim3 grid((unsigned)((nT * 64 + 255) / 256));
  switch (m) {
#define DBCSR_TILE_REM(S_)                                                                                                              \
  case S_:                                                                                                                              \
    hipLaunchKernelGGL((tile_remainder<S_, S_>), grid, dim3(256), 0, st, G, tdescs, entries, a_data, b_data, c_out, alpha);             \
    return check(hipGetLastError(), "tile_remainder", __FILE__, __LINE__);
    DBCSR_AMD_TILE_SIZES(DBCSR_TILE_REM)
#undef DBCSR_TILE_REM
    default: return 1;
  }
}

}  // namespace dbcsr_amd
