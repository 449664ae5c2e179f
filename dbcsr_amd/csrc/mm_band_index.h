// mm_band_index.h -- index kernels of the band dataflow (mm_band.h): per sub-tile the descriptors of its C blocks, per (tile, wave) ONE
// k-sorted product list whose entries carry what the ring protocol needs (sequence number and number of users of the B block, new-A and
// last-use flags), laid out in the order the workgroups sweep their tiles.  Built from the bitmaps of A (rows) and of B transposed
// (columns) with wave-wide prefix sums; integer work, bit-exact by construction, checked against the per-block product counts.  The host
// sequence is run_band_f64 in mm_engine.hip (the only file that includes this one).
#ifndef DBCSR_AMD_MM_BAND_INDEX_H
#define DBCSR_AMD_MM_BAND_INDEX_H

#include "mm_band.h"

namespace dbcsr_amd {

// tile T -> the workgroup that sweeps it (g = xcd * cu_per_xcd + cu) and its position in that workgroup's sweep
__device__ __forceinline__ void band_owner(const BandGeom& G, int T, int* g, int* i) {
  int x = 0;
#pragma unroll
  for (int q = 1; q < 8; ++q)
    if ((int64_t)T >= G.lo(q)) x = q;
  const int r = T - (int)G.lo(x);
  *g = x * G.cu_per_xcd + r % G.cu_per_xcd;
  *i = r / G.cu_per_xcd;
}

// one lane per (sub-tile, slot): offsets of the slot's C block, product count of the sub-tile.  Sub-tile (8 band + w, ct) holds the C
// blocks (rows[tr (waves band + w) + ti], cols[tc ct + tj]).
__global__ void __launch_bounds__(256) band_descs(BandGeom G, const int* __restrict__ rows, const int* __restrict__ cols,
                                                  const uint32_t* __restrict__ c_bm, const int* __restrict__ c_pre, const int* __restrict__ c_row_p,
                                                  int W, const Desc* __restrict__ descs, BandDesc* __restrict__ bd, int* __restrict__ sub_cnt) {
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;  // 16 lanes per sub-tile, one per slot
  const int s = threadIdx.x & 15;
  const int64_t nsub = (int64_t)G.waves * G.nBR * G.nBC;
  if (t >= nsub) return;
  const int tr = (int)(t / G.nBC), tc = (int)(t % G.nBC);
  int cnt = 0;
  int64_t c_off = -1, cin_off = -1;
  if (s < kBandSlots) {
    const int ri = G.tr * tr + s / G.tc, ci = G.tc * tc + s % G.tc;
    if (s < G.tr * G.tc && ri < G.nfr && ci < G.nfc) {
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
    bd[t].c_off[s] = c_off;
    bd[t].cin_off[s] = cin_off;
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 16);
  if (s == 0) sub_cnt[t] = cnt;
}

// One wavefront per (tile, wave).  Lane l of a trip looks at inner block k = k0 + l: which of the tile's 24 rows have A(i, k), which of
// its 3 columns have B(k, j).  FILL = false: counts (list length incl. the end-of-tile marker, B copies of the tile, products with an
// inner block of another size); FILL = true: the entries, at the offsets the scans of the counts gave.
template <bool FILL>
__global__ void __launch_bounds__(256)
band_lists(BandGeom G, const int* __restrict__ rows, const int* __restrict__ cols, int nbk, int Wk, const uint32_t* __restrict__ a_bm,
           const int* __restrict__ a_pre, const int* __restrict__ a_row_p, const int64_t* __restrict__ a_blk_p, const uint32_t* __restrict__ bt_bm,
           int W, const uint32_t* __restrict__ b_bm, const int* __restrict__ b_pre, const int* __restrict__ b_row_p,
           const int64_t* __restrict__ b_blk_p, const int* __restrict__ k_sizes, int K, int* __restrict__ cnt_list, int* __restrict__ cnt_b,
           int* __restrict__ cnt_rem, const int64_t* __restrict__ list_off, const int64_t* __restrict__ seq_off,
           const int64_t* __restrict__ rem_start, const int* __restrict__ sub_cnt, BandEntry* __restrict__ entries, BandRem* __restrict__ rem,
           int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int64_t u = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (u >= (int64_t)G.ntiles * G.waves) return;
  const int T = (int)(u / G.waves), w = (int)(u % G.waves);
  const int band_rows = G.waves * G.tr;
  const int band = T / G.nBC, ct = T % G.nBC;
  int g, it;
  band_owner(G, T, &g, &it);
  const int64_t pl = (int64_t)(g * G.waves + w) * G.max_i + it;  // this list, in processing order
  const int64_t ps = (int64_t)g * G.max_i + it;                     // this tile, in processing order
  const int64_t sub = (int64_t)(G.waves * band + w) * G.nBC + ct;
  int cj[kBandMaxT], ri[kBandMaxT];
#pragma unroll
  for (int q = 0; q < kBandMaxT; ++q) {
    cj[q] = (q < G.tc && G.tc * ct + q < G.nfc) ? cols[G.tc * ct + q] : -1;
    ri[q] = (q < G.tr && band_rows * band + G.tr * w + q < G.nfr) ? rows[band_rows * band + G.tr * w + q] : -1;
  }
  const unsigned trmask = (1u << G.tr) - 1u;
  int64_t pos = 0, rpos = 0;
  unsigned seq = 0;
  if (FILL) {
    pos = list_off[pl];
    rpos = rem_start[sub];
    seq = (unsigned)(seq_off[ps] - seq_off[(int64_t)g * G.max_i]);
  }
  int n_main = 0, n_rem = 0, n_b = 0;
  for (int k0 = 0; k0 < nbk; k0 += 64) {
    const int k = k0 + lane;
    const bool kin = k < nbk;
    unsigned am24 = 0, bm = 0;  // (am24: the rows of the whole tile, at most 32)
    for (int r = 0; r < band_rows; ++r) {
      const int rr = band_rows * band + r;
      const int row = rr < G.nfr ? rows[rr] : -1;  // (wave-uniform)
      const uint32_t word = (kin && row >= 0) ? a_bm[(size_t)row * Wk + (k >> 5)] : 0u;
      am24 |= ((word >> (k & 31)) & 1u) << r;
    }
#pragma unroll
    for (int q = 0; q < kBandMaxT; ++q) {
      const uint32_t word = (kin && cj[q] >= 0) ? bt_bm[(size_t)cj[q] * Wk + (k >> 5)] : 0u;
      bm |= ((word >> (k & 31)) & 1u) << q;
    }
    const unsigned amw = (am24 >> (G.tr * w)) & trmask;
    const int ks = kin ? k_sizes[k] : 0;
    const bool main = ks == K;
    const int nbl = (main && am24 != 0u) ? __popc(bm) : 0;  // B copies of the tile at this k
    const int np = __popc(amw) * __popc(bm);
    int inc_m = main ? np : 0, inc_r = main ? 0 : np, inc_b = nbl;
    const int my_m = inc_m, my_r = inc_r, my_b = inc_b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int tm = __shfl_up(inc_m, off, 64), tr = __shfl_up(inc_r, off, 64), tb = __shfl_up(inc_b, off, 64);
      if (lane >= off) inc_m += tm, inc_r += tr, inc_b += tb;
    }
    if (FILL && np > 0) {
      unsigned users = 0;
      for (int q = 0; q < G.waves; ++q) users += ((am24 >> (G.tr * q)) & trmask) != 0u;
      int64_t boff[kBandMaxT];
#pragma unroll
      for (int q = 0; q < kBandMaxT; ++q) {
        boff[q] = 0;
        if ((bm >> q) & 1u) {
          const int j = cj[q];
          const uint32_t bw = b_bm[(size_t)k * W + (j >> 5)];
          boff[q] = b_blk_p[b_row_p[k] + b_pre[(size_t)k * W + (j >> 5)] + __popc(bw & ((1u << (j & 31)) - 1u))];
        }
      }
      int64_t at = pos + (inc_m - my_m), rat = rpos + (inc_r - my_r);
      const unsigned sq = seq + (unsigned)(inc_b - my_b);
      const int last_ti = 31 - __clz((int)amw);
#pragma unroll
      for (int ti = 0; ti < kBandMaxT; ++ti) {
        if (!((amw >> ti) & 1u)) continue;
        const int row = ri[ti];
        const uint32_t aw = a_bm[(size_t)row * Wk + (k >> 5)];
        const int64_t a = a_blk_p[a_row_p[row] + a_pre[(size_t)row * Wk + (k >> 5)] + __popc(aw & ((1u << (k & 31)) - 1u))];
        bool first = true;
        unsigned jj = 0;
#pragma unroll
        for (int tj = 0; tj < kBandMaxT; ++tj) {
          if (!((bm >> tj) & 1u)) continue;
          const int64_t b = boff[tj];
          const uint32_t hi = ((uint32_t)(((uint64_t)a >> 32) & 0xffu) << 16) | ((uint32_t)(((uint64_t)b >> 32) & 0xffu) << 24);
          const uint32_t slot = (uint32_t)(G.tc * ti + tj);
          if (main) {
            BandEntry e;
            e.a_lo = (uint32_t)a;
            e.b_lo = (uint32_t)b;
            const uint32_t kq = (uint32_t)(k >> G.kshift);
            e.w = slot | (first ? kBandNewA : 0u) | (ti == last_ti ? kBandLastB : 0u) | ((kq & 255u) << 8) | hi;
            e.s = ((sq + jj) & 0x7fffffu) | (users << 23) | ((kq >> 8) << 28);
            entries[at++] = e;
          } else {
            BandRem e;
            e.a_lo = (uint32_t)a;
            e.b_lo = (uint32_t)b;
            e.w = slot | ((uint32_t)ks << 8) | hi;
            e.pad = 0;
            rem[rat++] = e;
          }
          first = false;
          ++jj;
        }
      }
    }
    const int tot_m = __shfl(inc_m, 63, 64), tot_r = __shfl(inc_r, 63, 64), tot_b = __shfl(inc_b, 63, 64);
    n_main += tot_m;
    n_rem += tot_r;
    n_b += tot_b;
    pos += tot_m;
    rpos += tot_r;
    seq += (unsigned)tot_b;
  }
  if (lane == 0) {
    if (FILL) {
      BandEntry e;
      e.a_lo = e.b_lo = e.s = 0u;
      e.w = kBandNop | kBandFlush;
      entries[pos] = e;  // end of the tile: the wave writes its C blocks
      if (n_main + n_rem != sub_cnt[sub]) atomicAdd(err, 1);  // the per-block product counts and the lists must agree
    } else {
      cnt_list[pl] = n_main + 1;
      cnt_rem[sub] = n_rem;
      if (w == 0) cnt_b[ps] = n_b;
    }
  }
}

// largest number of B copies any workgroup's sweep holds (the entries carry 24 bits of it)
__global__ void __launch_bounds__(256) band_max_seq(BandGeom G, int nwg, const int64_t* __restrict__ seq_off, int* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nwg) return;
  const int64_t n = seq_off[(int64_t)(g + 1) * G.max_i] - seq_off[(int64_t)g * G.max_i];
  atomicMax(out, (int)(n > 0x7fffffff ? 0x7fffffff : n));
}

}  // namespace dbcsr_amd
#endif
