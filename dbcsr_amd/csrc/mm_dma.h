// mm_dma.h -- fp64 block-product kernels whose operands reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds).
//
// Round-1 kernels staged every A and B block through VGPRs (raw buffer load -> 40 VGPRs -> 10 ds_write_b128 per
// 23^3 product); the VGPR->LDS transfer alone kept the kernel above 15 ms on BASELINE config 2 with no global load
// at all.  Here a wave owns a ring of S LDS slots (one slot = the A and the B block of one
// product, laid out as stored); the blocks of product p + S - 1 are requested with 1 KiB-per-instruction LDS-DMA
// pieces (no VGPR, no ds_write) while product p is multiplied, and the wave waits with a COUNTED s_waitcnt vmcnt
// for exactly the pieces of product p.  One wavefront per workgroup: no barriers anywhere, LDS occupancy is chosen
// by S alone (23^3: slot 8480 B; S = 2 -> 9 waves per CU, S = 3 -> 6, S = 4 -> 4).
//
// The DMA is issued from inline asm (the compiler would otherwise serialise every ds_read behind vmcnt(0) as soon
// as one LDS-DMA is in flight); rules followed (cdna_hip_programming.md section 5.7): M0 is written in the same
// statement that uses it, one wait state between the M0 write and the DMA, the partial last piece of a block runs
// under an EXEC mask set and restored inside the statement, and no compiler-visible store is in flight while a
// counted wait is used (stores and loads retire out of order with respect to each other on the vmcnt counter).
#ifndef DBCSR_AMD_MM_DMA_H
#define DBCSR_AMD_MM_DMA_H

#include "dma_lds.h"

namespace dbcsr_amd {

// One C block of exactly M x N whose products have inner dimension K (all compile-time), S-slot LDS-DMA ring.
template <int M, int N, int K, int S>
__device__ __forceinline__ void cblock_f64_dma(const Desc& d, const Entry* __restrict__ entries, const double* __restrict__ a_data,
                                               const double* __restrict__ b_data, double* __restrict__ c_out,
                                               const double* __restrict__ c_in, double alpha, double beta, const LaneMap& L, int lane,
                                               char* ring) {
  constexpr int MA = (M + 7) / 8, NC = (N + 7) / 8, KS = (K + 3) / 4;
  constexpr int ABYTES = M * K * 8, BBYTES = K * N * 8;
  constexpr int SA = (ABYTES + 15) & ~15, SB = (BBYTES + 15) & ~15, SLOT = SA + SB;
  constexpr int PIECES = (ABYTES + 1023) / 1024 + (BBYTES + 1023) / 1024;  // DMA instructions per product
  static_assert((S - 1) * PIECES < 64, "ring too deep for the vmcnt counter");
  double acc[MA][NC];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = 0.0;
  const Entry* e = entries + d.prod_start;
  const int cnt = d.prod_cnt;
  const int voff = lane * 16;
  const unsigned ring_lds = lds_offset_of(ring);
  // fragment offsets inside a slot (doubles): constant for the whole life of the wave
  int oa[MA], ob[NC], obt[NC];
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    int row = 8 * a + L.rowl;
    row = row < M ? row : M - 1;
    oa[a] = row + M * L.kq;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = 8 * c + L.coll;
    col = col < N ? col : N - 1;
    ob[c] = SA / 8 + L.kq + K * col;
    const int kt = 4 * (KS - 1) + L.kq;  // last k step when K is not a multiple of 4: lanes past the end read (0, col)
    obt[c] = SA / 8 + (kt < K ? kt : 0) + K * col;
  }
  const bool ktail_dead = (K & 3) != 0 && (4 * (KS - 1) + L.kq) >= K;  // this lane's k index is past the end in the last step
  // The product list is read ONCE with a vector load (lane l holds entry base + l) and handed out with v_readlane:
  // no scalar-load latency between two products, and no SMEM in flight next to the ds_reads (SMEM returns out of
  // order, so a pending s_load would turn every lgkmcnt wait of the fragment reads into lgkmcnt(0)).
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
  auto entry_at = [&](int i) {  // i wave-uniform, ebase <= i
    if (i - ebase >= 64) load_window(i);
    const int j = __builtin_amdgcn_readfirstlane(i - ebase);
    Entry en;
    en.a_lo = (uint32_t)__builtin_amdgcn_readlane((int)ev0, j);
    en.b_lo = (uint32_t)__builtin_amdgcn_readlane((int)ev1, j);
    en.w = (uint32_t)__builtin_amdgcn_readlane((int)ev2, j);
    return en;
  };
  // products with inner dimension K go through the ring, in list order; the others (tail block column of A) afterwards
  auto next_k = [&](int i) {
    while (i < cnt && entry_at(i).ks() != K) ++i;
    return i;
  };
  auto issue = [&](int i, int slot) {
    const Entry en = entry_at(i);
    const unsigned lds = ring_lds + (unsigned)slot * SLOT;
    dma_block<ABYTES>(a_data + en.a_off(), lds, voff);
    dma_block<BBYTES>(b_data + en.b_off(), lds + SA, voff);
  };
  auto multiply = [&](int slot) {
    const double* sl = reinterpret_cast<const double*>(ring + slot * SLOT);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      double av[MA], bv[NC];
#pragma unroll
      for (int a = 0; a < MA; ++a) {
        av[a] = sl[oa[a] + s * 4 * M];
        if (s == KS - 1 && (K & 3)) av[a] = ktail_dead ? 0.0 : av[a];
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) bv[c] = (s == KS - 1 && (K & 3)) ? sl[obt[c]] : sl[ob[c] + 4 * s];
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
    }
  };
  // ring indices: q[j] = list index of the product in flight j steps ahead of the one being multiplied (cnt: none)
  int q[S];
  q[0] = next_k(0);
#pragma unroll
  for (int j = 1; j < S; ++j) q[j] = q[j - 1] < cnt ? next_k(q[j - 1] + 1) : cnt;
#pragma unroll
  for (int j = 0; j < S - 1; ++j)
    if (q[j] < cnt) issue(q[j], j);
  int slot = 0;  // slot of q[0]
  while (q[0] < cnt) {
    int slot_new = slot + S - 1;
    slot_new = slot_new >= S ? slot_new - S : slot_new;
    if (q[S - 1] < cnt) issue(q[S - 1], slot_new);
    // pieces younger than product q[0]'s: those of the products behind it that exist
    int ahead = 0;
#pragma unroll
    for (int j = 1; j < S; ++j) ahead += q[j] < cnt ? 1 : 0;
    ahead = __builtin_amdgcn_readfirstlane(ahead);
    switch (ahead) {
      case 0: dma_wait<0>(); break;
      case 1: dma_wait<(S > 1 ? 1 : 0) * PIECES>(); break;
      case 2: dma_wait<(S > 2 ? 2 : 0) * PIECES>(); break;
      default: dma_wait<(S > 3 ? 3 : 0) * PIECES>(); break;
    }
    multiply(slot);
    // the slot just read is refilled one trip later at the earliest, after these reads have fed their MFMAs
#pragma unroll
    for (int j = 0; j < S - 1; ++j) q[j] = q[j + 1];
    q[S - 1] = q[S - 2] < cnt ? next_k(q[S - 2] + 1) : cnt;
    slot = slot + 1 == S ? 0 : slot + 1;
  }
  for (int p = 0; p < cnt; ++p) {
    const Entry ep = e[p];
    if (ep.ks() != K) block_product_f64<MA, NC, false>(acc, a_data + ep.a_off(), b_data + ep.b_off(), M, N, ep.ks(), L);
  }
  // C epilogue through LDS (slot 0; every DMA has landed): the block leaves as stored, in whole 1 KiB pieces, streaming hint
  constexpr int CC = (M * N * 8 + 1023) / 1024;
  static_assert(CC * 1024 <= S * SLOT, "C staging must fit the ring");
  double* lds_c = reinterpret_cast<double*>(ring);
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
      if (row < M && col < N) lds_c[row + M * col] = alpha * acc[a][c];
    }
  const bool has_in = d.cin_off >= 0;
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(c_out + d.c_off), 0, M * N * 8, 0x00020000);
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  if (has_in) {
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(c_in + d.cin_off), 0, M * N * 8, 0x00020000);
    u32x4 ci[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) ci[c] = __builtin_amdgcn_raw_buffer_load_b128(rsi, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      f64x2 v = *reinterpret_cast<const f64x2*>(ring + c * 1024 + voff);
      const f64x2 w = __builtin_bit_cast(f64x2, ci[c]);
      v[0] += beta * w[0];
      v[1] += beta * w[1];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsc, voff + c * 1024, 0, 2);
    }
  } else {
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(ring + c * 1024 + voff);
      __builtin_amdgcn_raw_buffer_store_b128(v, rsc, voff + c * 1024, 0, 2);
    }
  }
}

template <int M, int N, int K, int S>
struct DmaRing {
  static constexpr int SLOT = ((M * K * 8 + 15) & ~15) + ((K * N * 8 + 15) & ~15);
  static constexpr int CBYTES = ((M * N * 8 + 1023) / 1024) * 1024;
  static constexpr int BYTES = (S * SLOT > CBYTES ? S * SLOT : CBYTES) + 16;  // + 16: C staging reads whole 16-byte lanes
};

// One wavefront per workgroup, one C block per wavefront (position blockIdx -> order[], XCD-contiguous as the other kernels).
template <int M, int N, int K, int S>
__global__ void __launch_bounds__(64) mm_numeric_f64_dma(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                         const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                         double* __restrict__ c_out, const double* __restrict__ c_in, double alpha,
                                                         double beta, int skip_empty, const int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int pos = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t cb = order[pos];
  if (cb < 0 || cb >= nblk) return;
  const Desc d = descs[cb];
  if (skip_empty && d.prod_cnt == 0) return;
  const LaneMap L(lane);
  if (d.m == M && d.n == N) {
    cblock_f64_dma<M, N, K, S>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, L, lane, smem);
    return;
  }
  cblock_f64<4, 4>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, L, 0, 0);  // tail block row / column
}

}  // namespace dbcsr_amd
#endif
