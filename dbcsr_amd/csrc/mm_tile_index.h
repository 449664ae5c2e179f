// mm_tile_index.h -- index kernels of the tile dataflow (mm_tile.h): per sub-tile the descriptors of its C blocks and ONE
// k-sorted product list, built from the bitmaps of A (rows) and of B transposed (columns) with wave-wide prefix sums.  Integer work,
// bit-exact by construction; the host sequence is run_tile_f64 in mm_engine.hip (the only file that includes this one).
#ifndef DBCSR_AMD_MM_TILE_INDEX_H
#define DBCSR_AMD_MM_TILE_INDEX_H

#include "mm_tile.h"

namespace dbcsr_amd {

// ids of the block rows (columns) whose size is `size`, ascending: one wavefront, ballot compaction
__global__ void __launch_bounds__(64) tile_select(const int* __restrict__ sizes, int n, int size, int* __restrict__ ids, int cap) {
  const int lane = threadIdx.x;
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    const bool hit = i < n && sizes[i] == size;
    const unsigned long long m = __ballot(hit);
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (hit && pos < cap) ids[pos] = i;
    base += __popcll(m);
  }
}

// bit (col, row) of the transposed pattern for every block (row, col): one wavefront per block row
__global__ void __launch_bounds__(256) tile_bitmap_transposed(const int* __restrict__ row_p, const int* __restrict__ col_i, int nbr, int Wt,
                                                              uint32_t* __restrict__ bt) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= nbr) return;
  for (int b = row_p[row] + lane; b < row_p[row + 1]; b += 64) atomicOr(&bt[(size_t)col_i[b] * Wt + (row >> 5)], 1u << (row & 31));
}

// one lane per (sub-tile, slot): descriptor offsets of the slot's C block, product count of the sub-tile
__global__ void __launch_bounds__(256) tile_descs(TileGeom G, const int* __restrict__ rows, const int* __restrict__ cols,
                                                  const uint32_t* __restrict__ c_bm, const int* __restrict__ c_pre, const int* __restrict__ c_row_p,
                                                  int W, const Desc* __restrict__ descs, TileDesc* __restrict__ td, int* __restrict__ tile_cnt) {
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;  // 16 lanes per sub-tile, one per slot
  const int s = threadIdx.x & 15;
  if (t >= (int64_t)G.nTR * G.nTC) return;
  const int tr = (int)(t / G.nTC), tc = (int)(t % G.nTC);
  int cnt = 0;
  int64_t c_off = -1, cin_off = -1;
  if (s < G.tr * G.tc) {
    const int ri = G.tr * tr + s / G.tc, ci = G.tc * tc + s % G.tc;
    if (ri < G.nfr && ci < G.nfc) {
      const int i = rows[ri], j = cols[ci];
      const uint32_t cw = c_bm[(size_t)i * W + (j >> 5)];
      if ((cw >> (j & 31)) & 1u) {
        const int cb = c_row_p[i] + c_pre[(size_t)i * W + (j >> 5)] + __popc(cw & ((1u << (j & 31)) - 1u));
        const Desc d = descs[cb];
        c_off = d.c_off;
        cin_off = d.cin_off;
        cnt = d.prod_cnt;
      }
    }
    td[t].c_off[s] = c_off;
    td[t].cin_off[s] = cin_off;
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 16);
  if (s == 0) tile_cnt[t] = cnt;
}

// one wavefront per sub-tile: the k-sorted product list.  Lane l of a trip looks at inner block k = k0 + l: which of the tile's
// rows have A(i, k), which of its columns have B(k, j) (bitmap of B transposed), so popc x popc products; a wave-wide
// prefix sum gives every lane its place in the list.
__global__ void __launch_bounds__(256)
tile_lists(TileGeom G, const int* __restrict__ rows, const int* __restrict__ cols, int nbk, int Wk, const uint32_t* __restrict__ a_bm,
           const int* __restrict__ a_pre, const int* __restrict__ a_row_p, const int64_t* __restrict__ a_blk_p, const uint32_t* __restrict__ bt_bm,
           int W, const uint32_t* __restrict__ b_bm, const int* __restrict__ b_pre, const int* __restrict__ b_row_p,
           const int64_t* __restrict__ b_blk_p, const int* __restrict__ k_sizes, int K, const int64_t* __restrict__ tile_start,
           const int* __restrict__ tile_cnt, TileDesc* __restrict__ td, TileEntry* __restrict__ entries, int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (t >= (int64_t)G.nTR * G.nTC) return;
  const int tr = (int)(t / G.nTC), tc = (int)(t % G.nTC);
  int ri[kTileMaxT], cj[kTileMaxT];
#pragma unroll
  for (int q = 0; q < kTileMaxT; ++q) {
    ri[q] = (q < G.tr && G.tr * tr + q < G.nfr) ? rows[G.tr * tr + q] : -1;
    cj[q] = (q < G.tc && G.tc * tc + q < G.nfc) ? cols[G.tc * tc + q] : -1;
  }
  // slots that have a C block (a product only exists where C has a block: C's pattern contains the product's by construction,
  // except under retain_sparsity, where products outside C_in's pattern are dropped)
  unsigned cmask = 0;
#pragma unroll
  for (int s = 0; s < kTileMaxSlots; ++s)
    if (s < G.tr * G.tc && td[t].c_off[s] >= 0) cmask |= 1u << s;
  const int64_t start = tile_start[t];
  const int total = tile_cnt[t];
  int n_main = 0, n_rem = 0;
  for (int k0 = 0; k0 < nbk; k0 += 64) {
    const int k = k0 + lane;
    const bool kin = k < nbk;
    unsigned am = 0, bm = 0;
    uint32_t aw[kTileMaxT], bw[kTileMaxT];
#pragma unroll
    for (int q = 0; q < kTileMaxT; ++q) {
      aw[q] = (kin && ri[q] >= 0) ? a_bm[(size_t)ri[q] * Wk + (k >> 5)] : 0u;
      bw[q] = (kin && cj[q] >= 0) ? bt_bm[(size_t)cj[q] * Wk + (k >> 5)] : 0u;
      am |= ((aw[q] >> (k & 31)) & 1u) << q;
      bm |= ((bw[q] >> (k & 31)) & 1u) << q;
    }
    // products of this lane: slot tc * ti + tj for every (ti, tj) with both operands and a C block
    unsigned pm = 0;
#pragma unroll
    for (int ti = 0; ti < kTileMaxT; ++ti)
      if ((am >> ti) & 1u) pm |= bm << (G.tc * ti);
    pm &= cmask;
    if (!__ballot(pm != 0)) continue;
    const int ks = kin ? k_sizes[k] : 0;
    const bool main = ks == K;
    const int np = __popc(pm);
    int inc_m = main ? np : 0, inc_r = main ? 0 : np;
    const int my_m = inc_m, my_r = inc_r;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int tm = __shfl_up(inc_m, off, 64), trr = __shfl_up(inc_r, off, 64);
      if (lane >= off) inc_m += tm, inc_r += trr;
    }
    int pos_m = n_main + inc_m - my_m, pos_r = n_rem + inc_r - my_r;
    n_main += __shfl(inc_m, 63, 64);
    n_rem += __shfl(inc_r, 63, 64);
    if (pm) {
      int64_t aoff[kTileMaxT], boff[kTileMaxT];
#pragma unroll
      for (int q = 0; q < kTileMaxT; ++q) {
        aoff[q] = boff[q] = 0;
        if ((am >> q) & 1u)
          aoff[q] = a_blk_p[a_row_p[ri[q]] + a_pre[(size_t)ri[q] * Wk + (k >> 5)] + __popc(aw[q] & ((1u << (k & 31)) - 1u))];
        if ((bm >> q) & 1u) {
          const int j = cj[q];
          const uint32_t w = b_bm[(size_t)k * W + (j >> 5)];
          boff[q] = b_blk_p[b_row_p[k] + b_pre[(size_t)k * W + (j >> 5)] + __popc(w & ((1u << (j & 31)) - 1u))];
        }
      }
#pragma unroll
      for (int s = 0; s < kTileMaxSlots; ++s)
        if ((pm >> s) & 1u) {
          const int64_t a = aoff[s / G.tc], b = boff[s % G.tc];
          TileEntry e;
          e.a_lo = (uint32_t)a;
          e.b_lo = (uint32_t)b;
          e.w = (uint32_t)s | ((uint32_t)ks << 8) | ((uint32_t)(((uint64_t)a >> 32) & 0xffu) << 16) | ((uint32_t)(((uint64_t)b >> 32) & 0xffu) << 24);
          e.k = (uint32_t)k;
          const int64_t at = main ? start + pos_m++ : start + total - 1 - pos_r++;
          if (at >= start && at < start + total) entries[at] = e;
        }
    }
  }
  if (lane == 0) {
    td[t].list_start = start;
    td[t].n_main = n_main;
    td[t].n_rem = n_rem;
    if (n_main + n_rem != total) atomicAdd(err, 1);  // the per-block product counts and the tile lists must agree
  }
}

}  // namespace dbcsr_amd
#endif
