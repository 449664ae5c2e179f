// mm_symbolic.h -- symbolic phase: bitmaps, product counts and lists, on-the-fly filter, C index, launch order (plain and per (m, n) class)
// Part of the device-resident multiply engine: included by mm_engine.hip (one translation unit), in this order:
// mm_workspace.h, mm_symbolic.h, mm_numeric_f64.h, mm_numeric_f32.h, mm_aux.h.
#ifndef DBCSR_AMD_MM_SYMBOLIC_H
#define DBCSR_AMD_MM_SYMBOLIC_H

namespace dbcsr_amd {

// ----------------------------------------------------------------------------
// symbolic kernels
// ----------------------------------------------------------------------------

// one wavefront per block row: set bit (row, col) for every block
__global__ void __launch_bounds__(256) bitmap_from_index(const int* __restrict__ row_p, const int* __restrict__ col_i, int nbr, int W,
                                                         uint32_t* __restrict__ bm) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= nbr) return;
  for (int b = row_p[row] + lane; b < row_p[row + 1]; b += 64) {
    const int j = col_i[b];
    atomicOr(&bm[(size_t)row * W + (j >> 5)], 1u << (j & 31));
  }
}

// Product matrix with symmetry, index in canonical (checkerboard) form: the local multiply computes block (i, j) only when it
// is the stored one of the pair (i, j) / (j, i) (dbcsr_mm_csr.F:280-292, checker_tr of dbcsr_dist_operations.F:65-75): the
// diagonal, (i + j) even above it, (i + j) odd below it.  Bits of word w of row i that may receive products:
__device__ __forceinline__ uint32_t canonical_bits(int i, int w) {
  const uint32_t even = (i & 1) ? 0xAAAAAAAAu : 0x55555555u;      // columns j of this word with (i + j) even (32 w is even)
  const int d = i - 32 * w;                                       // position of the diagonal relative to the word
  const uint32_t upper = d <= 0 ? 0xFFFFFFFFu : (d >= 32 ? 0u : ~((1u << d) - 1u));  // columns j >= i
  return (even & upper) | (~even & ~upper);
}

// thread per (row i, word w)
__global__ void __launch_bounds__(256) c_bitmap(const int* __restrict__ a_row_p, const int* __restrict__ a_col_i,
                                                const uint32_t* __restrict__ b_bm, const uint32_t* __restrict__ cin_bm, int nbr, int W,
                                                int retain, int canonical, uint32_t* __restrict__ c_bm) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nbr * W) return;
  const int i = (int)(t / W), w = (int)(t % W);
  uint32_t v = 0u;
  if (!retain) {
    for (int ab = a_row_p[i]; ab < a_row_p[i + 1]; ++ab) v |= b_bm[(size_t)a_col_i[ab] * W + w];
    if (canonical) v &= canonical_bits(i, w);
  }
  c_bm[t] = v | (cin_bm ? cin_bm[t] : 0u);
}

// one wavefront per row: exclusive prefix of popcounts inside the row + row total
__global__ void __launch_bounds__(256) row_prefix(const uint32_t* __restrict__ bm, int nbr, int W, int* __restrict__ pre,
                                                  int* __restrict__ row_nnz) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= nbr) return;
  int carry = 0;
  for (int base = 0; base < W; base += 64) {
    const int w = base + lane;
    const int c = w < W ? __popc(bm[(size_t)row * W + w]) : 0;
    int inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (w < W) pre[(size_t)row * W + w] = carry + inc - c;
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0 && row_nnz) row_nnz[row] = carry;
}

// thread per (row i, word w): per C block product count, block size, flop
__global__ void __launch_bounds__(256) count_products(const int* __restrict__ a_row_p, const int* __restrict__ a_col_i,
                                                      const int* __restrict__ rs, const int* __restrict__ ks, const int* __restrict__ cs,
                                                      const uint32_t* __restrict__ b_bm, const uint32_t* __restrict__ c_bm,
                                                      const int* __restrict__ c_pre, const int* __restrict__ c_row_p, int nbr, int W,
                                                      int* __restrict__ prod_cnt, int* __restrict__ blk_nze,
                                                      unsigned long long* __restrict__ flop_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long flop = 0;
  if (t < (int64_t)nbr * W) {
    const int i = (int)(t / W), w = (int)(t % W);
    uint32_t v = c_bm[t];
    if (v) {
      const int m = rs[i];
      int cb = c_row_p[i] + c_pre[t];
      const int a0 = a_row_p[i], a1 = a_row_p[i + 1];
      while (v) {
        const int bit = __ffs(v) - 1;
        v &= v - 1;
        const int n = cs[32 * w + bit];
        int cnt = 0;
        long long ksum = 0;
        for (int ab = a0; ab < a1; ++ab) {
          const int k = a_col_i[ab];
          if ((b_bm[(size_t)k * W + w] >> bit) & 1u) {
            ++cnt;
            ksum += ks[k];
          }
        }
        prod_cnt[cb] = cnt;
        blk_nze[cb] = m * n;
        flop += 2ull * (unsigned long long)m * n * ksum;
        ++cb;
      }
    }
  }
  // block reduce, one atomic per workgroup
  __shared__ unsigned long long red[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) flop += __shfl_down(flop, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = flop;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long s = red[0] + red[1] + red[2] + red[3];
    if (s) atomicAdd(flop_out, s);
  }
}

// thread per (row i, word w): emit C index, descriptors and product lists
__global__ void __launch_bounds__(256)
fill_products(const int* __restrict__ a_row_p, const int* __restrict__ a_col_i, const int64_t* __restrict__ a_blk_p,
              const int* __restrict__ b_row_p, const int64_t* __restrict__ b_blk_p, const int* __restrict__ cin_row_p,
              const int64_t* __restrict__ cin_blk_p, const int* __restrict__ rs, const int* __restrict__ ks,
              const int* __restrict__ cs, const uint32_t* __restrict__ b_bm, const int* __restrict__ b_pre,
              const uint32_t* __restrict__ cin_bm, const int* __restrict__ cin_pre, const uint32_t* __restrict__ c_bm,
              const int* __restrict__ c_pre, const int* __restrict__ c_row_p, const int64_t* __restrict__ prod_start,
              const int64_t* __restrict__ c_blk_p_ws, int nbr, int W, int* __restrict__ c_col_i, int64_t* __restrict__ c_blk_p,
              Desc* __restrict__ descs, Entry* __restrict__ entries) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nbr * W) return;
  const int i = (int)(t / W), w = (int)(t % W);
  uint32_t v = c_bm[t];
  if (!v) return;
  const int m = rs[i];
  int cb = c_row_p[i] + c_pre[t];
  const int a0 = a_row_p[i], a1 = a_row_p[i + 1];
  const uint32_t cinw = cin_bm ? cin_bm[t] : 0u;
  while (v) {
    const int bit = __ffs(v) - 1;
    v &= v - 1;
    const uint32_t below = (1u << bit) - 1u;
    const int j = 32 * w + bit;
    int64_t p = prod_start[cb];
    int cnt = 0;
    for (int ab = a0; ab < a1; ++ab) {
      const int k = a_col_i[ab];
      const uint32_t bw = b_bm[(size_t)k * W + w];
      if ((bw >> bit) & 1u) {
        const int bidx = b_row_p[k] + b_pre[(size_t)k * W + w] + __popc(bw & below);
        entries[p + cnt] = Entry::make(a_blk_p[ab], b_blk_p[bidx], ks[k]);
        ++cnt;
      }
    }
    Desc d;
    d.c_off = c_blk_p_ws[cb];
    d.cin_off = -1;
    if ((cinw >> bit) & 1u) d.cin_off = cin_blk_p[cin_row_p[i] + cin_pre[t] + __popc(cinw & below)];
    d.prod_start = p;
    d.prod_cnt = cnt;
    d.m = (int16_t)m;
    d.n = (int16_t)cs[j];
    descs[cb] = d;
    c_col_i[cb] = j;
    c_blk_p[cb] = d.c_off;
    ++cb;
  }
}



// ---- on-the-fly filtering (dbcsr_mm_csr.F:276, dbcsr_mm_cannon.F:1040-1113) ------------------------
// A product A(i,k)*B(k,j) is skipped when ||A(i,k)||^2 * ||alpha B(k,j)||^2 < (eps / max(1, #blocks in A row i))^2,
// all in single precision as the reference (norms are fp32 values of fp64 sums).  a_norms == nullptr: no filter.
struct FilterArgs {
  const float* a_norms;
  const float* b_norms;
  float eps;
};

__device__ __forceinline__ float row_filter_eps(const FilterArgs& F, int nblks_in_a_row) {
  const float e = F.eps / (float)(nblks_in_a_row > 1 ? nblks_in_a_row : 1);
  return e * e;
}

// one wavefront per block row: norms[b] = (float) sum (scale * x)^2 over block b
template <typename T>
__global__ void __launch_bounds__(256) bcsr_block_norms(const int* __restrict__ row_p, const int* __restrict__ col_i,
                                                        const int64_t* __restrict__ blk_p, const T* __restrict__ data,
                                                        const int* __restrict__ rs, const int* __restrict__ cs, int nbr, int S, double scale,
                                                        float* __restrict__ norms, double* __restrict__ norms64) {
  // S waves share a block row (wave s takes the blocks b = s mod S): a long row is not one wave's serial stream
  const int lane = threadIdx.x & 63;
  const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int row = (int)(wv / S), sub = (int)(wv % S);
  if (row >= nbr) return;
  const int m = rs[row];
  for (int b = row_p[row] + sub; b < row_p[row + 1]; b += S) {
    const int ne = m * cs[col_i[b]];
    const T* d = data + blk_p[b];
    double s = 0.0;
    for (int e = lane; e < ne; e += 64) {
      const double x = scale * (double)d[e];
      s += x * x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
      if (norms) norms[b] = (float)s;
      if (norms64) norms64[b] = s;
    }
  }
}

// C pattern under filtering: one lane per (row i, column j) candidate, bit set iff C_in has the block or at
// least one product survives the filter (a new C block is only created by a product that is executed)
__global__ void __launch_bounds__(256) c_bitmap_filtered(const int* __restrict__ a_row_p, const int* __restrict__ a_col_i,
                                                         const int* __restrict__ b_row_p, const uint32_t* __restrict__ b_bm,
                                                         const int* __restrict__ b_pre, const uint32_t* __restrict__ cin_bm, int nbr,
                                                         int nbc, int W, int nJ, int retain, int canonical, FilterArgs F,
                                                         uint32_t* __restrict__ c_bm) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wv >= (int64_t)nbr * nJ) return;
  const int i = (int)(wv / nJ), jb = (int)(wv % nJ);
  const int j = jb * 64 + lane, w = j >> 5, bit = j & 31;
  bool any = false;
  if (!retain && j < nbc && !(canonical && !((canonical_bits(i, w) >> bit) & 1u))) {
    const int a0 = a_row_p[i], a1 = a_row_p[i + 1];
    const float reps = row_filter_eps(F, a1 - a0);
    const uint32_t below = (1u << bit) - 1u;
    for (int ab = a0; ab < a1; ++ab) {
      const int k = a_col_i[ab];
      const uint32_t bw = b_bm[(size_t)k * W + w];
      if ((bw >> bit) & 1u) {
        const int bidx = b_row_p[k] + b_pre[(size_t)k * W + w] + __popc(bw & below);
        if (!(F.a_norms[ab] * F.b_norms[bidx] < reps)) any = true;
      }
    }
  }
  const unsigned long long mask = __ballot(any);
  if (lane == 0) {
    const int w0 = 2 * jb;
    c_bm[(size_t)i * W + w0] = (uint32_t)mask | (cin_bm ? cin_bm[(size_t)i * W + w0] : 0u);
    if (w0 + 1 < W) c_bm[(size_t)i * W + w0 + 1] = (uint32_t)(mask >> 32) | (cin_bm ? cin_bm[(size_t)i * W + w0 + 1] : 0u);
  }
}


// ---- dense-grid variants: one lane per (row i, column j) candidate ------------
// The per-word kernels above expose only nbr*W threads, each walking up to 32 C
// blocks x |A-row| serially (v1 profile: 1.6 + 4.0 ms for config 2).  When C is not
// extremely sparse it is much faster to give every candidate (i, j) its own lane:
// a wavefront covers 64 consecutive columns of one row, so the walk over A's row
// is wave-uniform (scalar loads) and the B bitmap words are two broadcast loads.
__global__ void __launch_bounds__(256) count_products_grid(const int* __restrict__ a_row_p, const int* __restrict__ a_col_i,
                                                           const int* __restrict__ rs, const int* __restrict__ ks,
                                                           const int* __restrict__ cs, const uint32_t* __restrict__ b_bm,
                                                           const uint32_t* __restrict__ c_bm, const int* __restrict__ c_pre,
                                                           const int* __restrict__ c_row_p, int nbr, int nbc, int W, int nJ,
                                                           int* __restrict__ prod_cnt, int* __restrict__ blk_nze,
                                                           unsigned long long* __restrict__ flop_out, const int* __restrict__ b_row_p,
                                                           const int* __restrict__ b_pre, FilterArgs F) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  unsigned long long flop = 0;
  if (wv < (int64_t)nbr * nJ) {
    const int i = (int)(wv / nJ), jb = (int)(wv % nJ);
    const int j = jb * 64 + lane, w = j >> 5, bit = j & 31;
    const uint32_t cw = w < W ? c_bm[(size_t)i * W + w] : 0u;
    const bool present = (cw >> bit) & 1u;
    if (__ballot(present)) {
      const int a0 = a_row_p[i], a1 = a_row_p[i + 1];
      const float reps = F.a_norms ? row_filter_eps(F, a1 - a0) : 0.0f;
      int cnt = 0;
      long long ksum = 0;
      for (int ab = a0; ab < a1; ++ab) {
        const int k = a_col_i[ab];
        const uint32_t bw = w < W ? b_bm[(size_t)k * W + w] : 0u;
        if ((bw >> bit) & 1u) {
          if (F.a_norms) {
            const int bidx = b_row_p[k] + b_pre[(size_t)k * W + w] + __popc(bw & ((1u << bit) - 1u));
            if (F.a_norms[ab] * F.b_norms[bidx] < reps) continue;
          }
          ++cnt;
          ksum += ks[k];
        }
      }
      if (present) {
        const int cb = c_row_p[i] + c_pre[(size_t)i * W + w] + __popc(cw & ((1u << bit) - 1u));
        const int m = rs[i], n = cs[j];
        prod_cnt[cb] = cnt;
        blk_nze[cb] = m * n;
        flop = 2ull * (unsigned long long)m * n * (unsigned long long)ksum;
      }
    }
  }
  __shared__ unsigned long long red[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) flop += __shfl_down(flop, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = flop;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(flop_out, t);
  }
}

__global__ void __launch_bounds__(256)
fill_products_grid(const int* __restrict__ a_row_p, const int* __restrict__ a_col_i, const int64_t* __restrict__ a_blk_p,
                   const int* __restrict__ b_row_p, const int64_t* __restrict__ b_blk_p, const int* __restrict__ cin_row_p,
                   const int64_t* __restrict__ cin_blk_p, const int* __restrict__ rs, const int* __restrict__ ks,
                   const int* __restrict__ cs, const uint32_t* __restrict__ b_bm, const int* __restrict__ b_pre,
                   const uint32_t* __restrict__ cin_bm, const int* __restrict__ cin_pre, const uint32_t* __restrict__ c_bm,
                   const int* __restrict__ c_pre, const int* __restrict__ c_row_p, const int64_t* __restrict__ prod_start,
                   const int64_t* __restrict__ c_blk_p_ws, int nbr, int nbc, int W, int nJ, int* __restrict__ c_col_i,
                   int64_t* __restrict__ c_blk_p, Desc* __restrict__ descs, Entry* __restrict__ entries, FilterArgs F) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wv >= (int64_t)nbr * nJ) return;
  const int i = (int)(wv / nJ), jb = (int)(wv % nJ);
  const int j = jb * 64 + lane, w = j >> 5, bit = j & 31;
  const uint32_t cw = w < W ? c_bm[(size_t)i * W + w] : 0u;
  const bool present = (cw >> bit) & 1u;
  if (!__ballot(present)) return;
  const uint32_t below = (1u << bit) - 1u;
  const int cb = present ? c_row_p[i] + c_pre[(size_t)i * W + w] + __popc(cw & below) : 0;
  const int64_t p0 = present ? prod_start[cb] : 0;
  const int a0 = a_row_p[i], a1 = a_row_p[i + 1];
  const float reps = F.a_norms ? row_filter_eps(F, a1 - a0) : 0.0f;
  int cnt = 0;
  // A's row is walked in chunks of 64 blocks: first a cheap pass that only tests the B bitmap and records the
  // hits of this lane in a 64-bit mask, then the expensive part (index look-ups, entry store) runs per HIT -- about
  // fill x 64 trips per chunk instead of 64 (v4: 0.85 ms for config 2 with the one-pass loop).
  for (int base = a0; base < a1; base += 64) {
    const int top = min(base + 64, a1);
    unsigned long long hits = 0ull;
    for (int ab = base; ab < top; ++ab) {
      const int k = a_col_i[ab];
      const uint32_t bw = w < W ? b_bm[(size_t)k * W + w] : 0u;
      if (present && ((bw >> bit) & 1u)) hits |= 1ull << (ab - base);
    }
    while (hits) {
      const int ab = base + __ffsll((long long)hits) - 1;
      hits &= hits - 1;
      const int k = a_col_i[ab];
      const uint32_t bw = b_bm[(size_t)k * W + w];
      const int bidx = b_row_p[k] + b_pre[(size_t)k * W + w] + __popc(bw & below);
      if (F.a_norms && F.a_norms[ab] * F.b_norms[bidx] < reps) continue;
      entries[p0 + cnt] = Entry::make(a_blk_p[ab], b_blk_p[bidx], ks[k]);
      ++cnt;
    }
  }
  if (present) {
    Desc d;
    d.c_off = c_blk_p_ws[cb];
    d.cin_off = -1;
    if (cin_bm) {
      const uint32_t cinw = cin_bm[(size_t)i * W + w];
      if ((cinw >> bit) & 1u) d.cin_off = cin_blk_p[cin_row_p[i] + cin_pre[(size_t)i * W + w] + __popc(cinw & below)];
    }
    d.prod_start = p0;
    d.prod_cnt = cnt;
    d.m = (int16_t)rs[i];
    d.n = (int16_t)cs[j];
    descs[cb] = d;
    c_col_i[cb] = j;
    c_blk_p[cb] = d.c_off;
  }
}


// C pattern under filtering, product-driven (sparse C): one wave per block row i ORs the bit of every product that passes the
// on-the-fly filter into row i of c_bm, which starts as C_in's pattern (the candidate-driven c_bitmap_filtered tests
// nbr x nbc x |A row| combinations)
__global__ void __launch_bounds__(256) c_bitmap_rows_filtered(const int* __restrict__ a_row_p, const int* __restrict__ a_col_i,
                                                              const int* __restrict__ b_row_p, const int* __restrict__ b_col_i, int nbr, int W,
                                                              int canonical, FilterArgs F, uint32_t* __restrict__ c_bm) {
  const int lane = threadIdx.x & 63;
  const int i = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (i >= nbr) return;
  const float reps = row_filter_eps(F, a_row_p[i + 1] - a_row_p[i]);
  for (int ab = a_row_p[i]; ab < a_row_p[i + 1]; ++ab) {
    const int k = a_col_i[ab];
    for (int bb = b_row_p[k] + lane; bb < b_row_p[k + 1]; bb += 64) {
      if (F.a_norms[ab] * F.b_norms[bb] < reps) continue;
      const int j = b_col_i[bb], w = j >> 5, bit = j & 31;
      if (canonical && !((canonical_bits(i, w) >> bit) & 1u)) continue;
      atomicOr(&c_bm[(size_t)i * W + w], 1u << bit);
    }
  }
}

// ---- product-driven variants for a sparse C (BASELINE config 4: C 43 % full, 1.3 products per C block) --------------------------
// The grid kernels test every (row, column) candidate against every block of A's row: nbr x nbc x |A row| bitmap tests
// (config 4: 1.9 G for 18.6 M products; count 2.0 ms + fill 2.7 ms = 15 % of the multiply).  Here ONE WAVE owns a block row i of A
// (hence of C) and walks its blocks A(i, k) in ascending k; the lanes take the blocks B(k, j) of row k, look up the C block by
// bitmap rank and bump its counter.  Work is proportional to the number of products.  Inside a step all lanes hit different C
// blocks, steps are sequential and no other wave touches row i, so the list slots handed out by the atomic in the fill pass
// follow ascending k: same lists as the other kernels.
__global__ void __launch_bounds__(256) count_products_rows(const int* __restrict__ a_row_p, const int* __restrict__ a_col_i,
                                                           const int* __restrict__ rs, const int* __restrict__ ks, const int* __restrict__ cs,
                                                           const int* __restrict__ b_row_p, const int* __restrict__ b_col_i,
                                                           const uint32_t* __restrict__ c_bm, const int* __restrict__ c_pre,
                                                           const int* __restrict__ c_row_p, int nbr, int W, int* __restrict__ prod_cnt,
                                                           unsigned long long* __restrict__ flop_out, FilterArgs F) {
  const int lane = threadIdx.x & 63;
  const int i = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  unsigned long long flop = 0;
  if (i < nbr) {
    const unsigned long long m = (unsigned long long)rs[i];
    const float reps = F.a_norms ? row_filter_eps(F, a_row_p[i + 1] - a_row_p[i]) : 0.0f;
    for (int ab = a_row_p[i]; ab < a_row_p[i + 1]; ++ab) {
      const int k = a_col_i[ab];
      const unsigned long long kk = (unsigned long long)ks[k];
      for (int bb = b_row_p[k] + lane; bb < b_row_p[k + 1]; bb += 64) {
        if (F.a_norms && F.a_norms[ab] * F.b_norms[bb] < reps) continue;  // on-the-fly filter
        const int j = b_col_i[bb], w = j >> 5, bit = j & 31;
        const uint32_t cw = c_bm[(size_t)i * W + w];
        if (!((cw >> bit) & 1u)) continue;  // retain_sparsity / product matrix with symmetry: no such C block
        const int cb = c_row_p[i] + c_pre[(size_t)i * W + w] + __popc(cw & ((1u << bit) - 1u));
        atomicAdd(&prod_cnt[cb], 1);
        flop += 2ull * m * (unsigned long long)cs[j] * kk;
      }
    }
  }
  __shared__ unsigned long long red[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) flop += __shfl_down(flop, off, 64);
  if (lane == 0) red[threadIdx.x >> 6] = flop;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(flop_out, t);
  }
}

__global__ void __launch_bounds__(256) fill_products_rows(const int* __restrict__ a_row_p, const int* __restrict__ a_col_i,
                                                          const int64_t* __restrict__ a_blk_p, const int* __restrict__ ks,
                                                          const int* __restrict__ b_row_p, const int* __restrict__ b_col_i,
                                                          const int64_t* __restrict__ b_blk_p, const uint32_t* __restrict__ c_bm,
                                                          const int* __restrict__ c_pre, const int* __restrict__ c_row_p,
                                                          const int64_t* __restrict__ prod_start, int nbr, int W, int* __restrict__ fill_cnt,
                                                          Entry* __restrict__ entries, FilterArgs F) {
  const int lane = threadIdx.x & 63;
  const int i = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (i >= nbr) return;
  const float reps = F.a_norms ? row_filter_eps(F, a_row_p[i + 1] - a_row_p[i]) : 0.0f;
  for (int ab = a_row_p[i]; ab < a_row_p[i + 1]; ++ab) {
    const int k = a_col_i[ab];
    const int kk = ks[k];
    const int64_t a_off = a_blk_p[ab];
    for (int bb = b_row_p[k] + lane; bb < b_row_p[k + 1]; bb += 64) {
      if (F.a_norms && F.a_norms[ab] * F.b_norms[bb] < reps) continue;
      const int j = b_col_i[bb], w = j >> 5, bit = j & 31;
      const uint32_t cw = c_bm[(size_t)i * W + w];
      if (!((cw >> bit) & 1u)) continue;
      const int cb = c_row_p[i] + c_pre[(size_t)i * W + w] + __popc(cw & ((1u << bit) - 1u));
      const int slot = atomicAdd(&fill_cnt[cb], 1);
      entries[prod_start[cb] + slot] = Entry::make(a_off, b_blk_p[bb], kk);
    }
  }
}

// thread per (row, bitmap word): element counts of the C blocks in index order (count pass of the rows variant) ...
__global__ void __launch_bounds__(256) block_sizes_rows(const uint32_t* __restrict__ c_bm, const int* __restrict__ c_pre,
                                                        const int* __restrict__ c_row_p, const int* __restrict__ rs, const int* __restrict__ cs,
                                                        int nbr, int W, int* __restrict__ blk_nze) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nbr * W) return;
  const int i = (int)(t / W), w = (int)(t % W);
  uint32_t v = c_bm[t];
  int cb = c_row_p[i] + c_pre[t];
  while (v) {
    const int bit = __ffs(v) - 1;
    v &= v - 1;
    blk_nze[cb++] = rs[i] * cs[32 * w + bit];
  }
}

// ... and their descriptors / index entries (fill pass)
__global__ void __launch_bounds__(256)
finish_descs_rows(const int* __restrict__ cin_row_p, const int64_t* __restrict__ cin_blk_p, const int* __restrict__ rs,
                  const int* __restrict__ cs, const uint32_t* __restrict__ cin_bm, const int* __restrict__ cin_pre,
                  const uint32_t* __restrict__ c_bm, const int* __restrict__ c_pre, const int* __restrict__ c_row_p,
                  const int64_t* __restrict__ c_blk_p_ws, const int64_t* __restrict__ prod_start, const int* __restrict__ prod_cnt, int nbr,
                  int W, int* __restrict__ c_col_i, int64_t* __restrict__ c_blk_p, Desc* __restrict__ descs) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nbr * W) return;
  const int i = (int)(t / W), w = (int)(t % W);
  uint32_t v = c_bm[t];
  if (!v) return;
  int cb = c_row_p[i] + c_pre[t];
  const uint32_t cinw = cin_bm ? cin_bm[t] : 0u;
  while (v) {
    const int bit = __ffs(v) - 1;
    v &= v - 1;
    const uint32_t below = (1u << bit) - 1u;
    const int j = 32 * w + bit;
    Desc d;
    d.c_off = c_blk_p_ws[cb];
    d.cin_off = -1;
    if ((cinw >> bit) & 1u) d.cin_off = cin_blk_p[cin_row_p[i] + cin_pre[t] + __popc(cinw & below)];
    d.prod_start = prod_start[cb];
    d.prod_cnt = prod_cnt[cb];
    d.m = (int16_t)rs[i];
    d.n = (int16_t)cs[j];
    descs[cb] = d;
    c_col_i[cb] = j;
    c_blk_p[cb] = d.c_off;
    ++cb;
  }
}

// The same, one LANE per (row, column) candidate (a wave covers 64 consecutive columns of a row): the blocks of a wave are consecutive in C's index, so the
// descriptors, column indices and offsets leave in whole cache lines -- finish_descs_rows gives a thread 32 candidates and its neighbours write 14 descriptors apart
// (config 4's shape, 14.3 M C blocks: 0.99 ms; round 6, session r06_60).
__global__ void __launch_bounds__(256)
finish_descs_grid(const int* __restrict__ cin_row_p, const int64_t* __restrict__ cin_blk_p, const int* __restrict__ rs,
                  const int* __restrict__ cs, const uint32_t* __restrict__ cin_bm, const int* __restrict__ cin_pre,
                  const uint32_t* __restrict__ c_bm, const int* __restrict__ c_pre, const int* __restrict__ c_row_p,
                  const int64_t* __restrict__ c_blk_p_ws, const int64_t* __restrict__ prod_start, const int* __restrict__ prod_cnt, int nbr,
                  int W, int nJ, int* __restrict__ c_col_i, int64_t* __restrict__ c_blk_p, Desc* __restrict__ descs) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wv >= (int64_t)nbr * nJ) return;
  const int i = (int)(wv / nJ), jb = (int)(wv % nJ);
  const int j = jb * 64 + lane, w = j >> 5, bit = j & 31;
  if (w >= W) return;
  const size_t t = (size_t)i * W + w;
  const uint32_t cw = c_bm[t];
  if (!((cw >> bit) & 1u)) return;
  const uint32_t below = (1u << bit) - 1u;
  const int cb = c_row_p[i] + c_pre[t] + __popc(cw & below);
  Desc d;
  d.c_off = c_blk_p_ws[cb];
  d.cin_off = -1;
  if (cin_bm) {
    const uint32_t cinw = cin_bm[t];
    if ((cinw >> bit) & 1u) d.cin_off = cin_blk_p[cin_row_p[i] + cin_pre[t] + __popc(cinw & below)];
  }
  d.prod_start = prod_start[cb];
  d.prod_cnt = prod_cnt[cb];
  d.m = (int16_t)rs[i];
  d.n = (int16_t)cs[j];
  descs[cb] = d;
  c_col_i[cb] = j;
  c_blk_p[cb] = d.c_off;
}

// ---- C structure only (multi-tick / Cannon use): emit the sorted index of the pattern
// computed by the symbolic phase and describe where each block's initial value comes from.
__global__ void __launch_bounds__(256)
emit_index(const int* __restrict__ cin_row_p, const int64_t* __restrict__ cin_blk_p, const int* __restrict__ rs,
           const int* __restrict__ cs, const uint32_t* __restrict__ cin_bm, const int* __restrict__ cin_pre,
           const uint32_t* __restrict__ c_bm, const int* __restrict__ c_pre, const int* __restrict__ c_row_p,
           const int64_t* __restrict__ c_blk_p_ws, int nbr, int W, int* __restrict__ c_col_i, int64_t* __restrict__ c_blk_p,
           Desc* __restrict__ descs) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nbr * W) return;
  const int i = (int)(t / W), w = (int)(t % W);
  uint32_t v = c_bm[t];
  if (!v) return;
  int cb = c_row_p[i] + c_pre[t];
  const uint32_t cinw = cin_bm ? cin_bm[t] : 0u;
  while (v) {
    const int bit = __ffs(v) - 1;
    v &= v - 1;
    const uint32_t below = (1u << bit) - 1u;
    const int j = 32 * w + bit;
    Desc d;
    d.c_off = c_blk_p_ws[cb];
    d.cin_off = -1;
    if ((cinw >> bit) & 1u) d.cin_off = cin_blk_p[cin_row_p[i] + cin_pre[t] + __popc(cinw & below)];
    d.prod_start = 0;
    d.prod_cnt = 0;
    d.m = (int16_t)rs[i];
    d.n = (int16_t)cs[j];
    descs[cb] = d;
    c_col_i[cb] = j;
    c_blk_p[cb] = d.c_off;
    ++cb;
  }
}

// one wavefront per C block: C_out = beta * C_in where the block existed, 0 elsewhere
template <typename T>
__global__ void __launch_bounds__(256) init_c_blocks(const Desc* __restrict__ descs, int64_t nblk, T* __restrict__ c_out,
                                                     const T* __restrict__ c_in, T beta) {
  const int lane = threadIdx.x & 63;
  const int64_t cb = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (cb >= nblk) return;
  const Desc d = descs[cb];
  const int ne = (int)d.m * (int)d.n;
  T* C = c_out + d.c_off;
  if (d.cin_off >= 0) {
    const T* Ci = c_in + d.cin_off;
    for (int e = lane; e < ne; e += 64) C[e] = beta * Ci[e];
  } else {
    for (int e = lane; e < ne; e += 64) C[e] = (T)0;
  }
}

// ----------------------------------------------------------------------------
// processing order of the C blocks (speed only; results do not depend on it)
//
// v2 measurement (profiles/r01_v2_*): with C swept row by row every B block is
// fetched from HBM once per A block-row that needs it (L2 miss rate 50 %,
// ~127 GB of HBM reads for 1.7 GB of operands, kernel HBM-bound at 5.5 TB/s).
// Order used instead: C is swept in COLUMN PANELS narrow enough that the B panel
// (all rows x panel columns) stays resident in the 256 MB Infinity Cache; inside
// a panel, block row i belongs to XCD (i mod 8), so its A block-row is fetched
// into exactly one private L2 and reused by all C blocks of that row-panel.
// order[] holds, for each XCD, its (panel-major, row-minor) list of C block
// indices, padded with -1 to a common length so that the contiguous workgroup
// ranges xcd_remap() hands to each XCD coincide with these lists.
// ----------------------------------------------------------------------------
// A key is (XCD x, panel p, row group g): the RG block rows i = 8 (g RG + t) + x, t < RG, restricted to panel p.
// The rows of a group are walked TOGETHER, column by column, so that a B block fetched for C(i,j) is still in
// L2 when C(i',j) of another row of the group needs it (142 (1 - 0.9^RG) distinct B blocks per column instead
// of 14.2 RG); RG is chosen so that the group's A block-rows fit the XCD's 4 MB L2 together.
// key = (x * NP + p) * NG + g ; cnt[key] = number of C blocks of the group inside the panel
__device__ __forceinline__ int panel_rank(const int* __restrict__ c_pre, const int* __restrict__ row_nnz, int i, int W, int w) {
  return w < W ? c_pre[(size_t)i * W + w] : row_nnz[i];  // C blocks of row i left of bitmap word w
}

__global__ void __launch_bounds__(256) order_count(const int* __restrict__ c_pre, const int* __restrict__ row_nnz, int nbr, int W, int PW,
                                                   int NP, int NG, int RG, int* __restrict__ cnt) {
  const int key = blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= 8 * NP * NG) return;
  const int g = key % NG, p = (key / NG) % NP, x = key / (NG * NP);
  int c = 0;
  for (int t = 0; t < RG; ++t) {
    const int i = 8 * (g * RG + t) + x;
    if (i < nbr) c += panel_rank(c_pre, row_nnz, i, W, (p + 1) * PW) - panel_rank(c_pre, row_nnz, i, W, p * PW);
  }
  cnt[key] = c;
}

// thread per (row i, bitmap word w): position of each C block in the order of its XCD
__global__ void __launch_bounds__(256) order_fill(const uint32_t* __restrict__ c_bm, const int* __restrict__ c_pre,
                                                  const int* __restrict__ row_nnz, const int* __restrict__ c_row_p,
                                                  const int64_t* __restrict__ base, int nbr, int W, int PW, int NP, int NG, int RG,
                                                  int64_t len, int* __restrict__ order) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (int64_t)nbr * W) return;
  const int i = (int)(tid / W), w = (int)(tid % W);
  uint32_t v = c_bm[tid];
  if (!v) return;
  const int x = i & 7, r = i >> 3, g = r / RG, t = r % RG, p = w / PW;
  const int key = (x * NP + p) * NG + g;
  const int64_t dst0 = (int64_t)x * len + (base[key] - base[(size_t)x * NP * NG]);
  const int w0 = p * PW;
  int cb = c_row_p[i] + c_pre[tid];
  while (v) {
    const int bit = __ffs(v) - 1;
    v &= v - 1;
    const uint32_t below = (1u << bit) - 1u;
    // blocks of the group that come before (column j, row slot t): all blocks of the group's rows with a smaller
    // column (inside the panel), plus the rows before t that own column j
    int before = 0;
    for (int tt = 0; tt < RG; ++tt) {
      const int ii = 8 * (g * RG + tt) + x;
      if (ii >= nbr) break;
      const uint32_t ww = c_bm[(size_t)ii * W + w];
      before += c_pre[(size_t)ii * W + w] + __popc(ww & below) - panel_rank(c_pre, row_nnz, ii, W, w0);
      if (tt < t) before += (ww >> bit) & 1u;
    }
    order[dst0 + before] = cb;
    ++cb;
  }
}

// per-XCD totals -> common padded length (multiple of 4), written to out[0]
__global__ void order_len(const int64_t* __restrict__ base, int64_t total, int NP, int NG, int64_t* __restrict__ out) {
  int64_t mx = 0;
  for (int x = 0; x < 8; ++x) {
    const int64_t b0 = base[(size_t)x * NP * NG];
    const int64_t b1 = x < 7 ? base[(size_t)(x + 1) * NP * NG] : total;
    mx = b1 - b0 > mx ? b1 - b0 : mx;
  }
  out[0] = (mx + 3) & ~(int64_t)3;
}


// ----------------------------------------------------------------------------
// (m, n) classes of C blocks (mixed block sizes): order[] in one segment per class
//
// The reference sorts block products into homogeneous stacks by the three most common sizes of each dimension
// (map_most_common, src/dist/dbcsr_dist_util.F:753-812; stack_map in dbcsr_mm_csr.F:497-525) and runs each stack on the
// kernel compiled for its (m, n, k).  Here a C block is the unit of work, so C blocks are bucketed by (m, n): class
// c = 3 * rank(m) + rank(n) for the three most common row and column block sizes (ranks 0..2), class 9 = everything else.
// order[] becomes ten segments, each laid out like the single list of the other kernels (eight XCD streams padded to a
// common length, column panels, the rows of a row class dealt to the XCDs in turn: class_row_deal), and each segment is one launch of the kernel for its class.
// ----------------------------------------------------------------------------
constexpr int kNumClasses = 10;

__global__ void __launch_bounds__(256) class_ids(const int* __restrict__ sizes, int n, int s0, int s1, int s2, unsigned char* __restrict__ cls) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = sizes[i];
  cls[i] = (unsigned char)(s == s0 ? 0 : (s == s1 ? 1 : (s == s2 ? 2 : 3)));
}

// ncls_bm[q * W + w]: bit j of word w set iff column 32 w + j has class q (q = 0..3)
__global__ void __launch_bounds__(256) class_col_bitmaps(const unsigned char* __restrict__ ncls, int nbc, int W, uint32_t* __restrict__ bm) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  uint32_t m[4] = {0u, 0u, 0u, 0u};
  for (int b = 0; b < 32; ++b) {
    const int j = 32 * w + b;
    if (j < nbc) m[ncls[j]] |= 1u << b;
  }
  for (int q = 0; q < 4; ++q) bm[(size_t)q * W + w] = m[q];
}

// columns of row i (class rc) that belong to class `cls`, as a mask on bitmap word w
__device__ __forceinline__ uint32_t class_mask(int cls, int rc, const uint32_t* __restrict__ ncls_bm, int W, int w) {
  if (cls < 9) return rc == cls / 3 ? ncls_bm[(size_t)(cls % 3) * W + w] : 0u;
  return rc == 3 ? 0xffffffffu : ncls_bm[(size_t)3 * W + w];
}

// Rows dealt to the XCDs class by class (round 6): with "row i on XCD i mod 8" a size pattern whose period divides 8 -- two atom kinds alternating: 5, 13,
// 5, 13 ... -- puts every row class on HALF of the XCDs, and each class launch then runs on half of the chip (measured: 9.4 ms where the run-time-size
// kernel, one launch over all rows, takes 7.2).  The rows are numbered again, class after class in row order (vpos[i] = virtual row of row i,
// vrow[] its inverse); virtual row j goes to XCD j mod 8, so the rows of every class are spread over all eight.  One workgroup.
__global__ void __launch_bounds__(256) class_row_deal(const unsigned char* __restrict__ rowcls, int nbr, int* __restrict__ vpos, int* __restrict__ vrow) {
  __shared__ int cnt[256][4];
  __shared__ int cls_off[4];
  const int t = threadIdx.x, chunk = (nbr + 255) / 256, i0 = t * chunk, i1 = min(nbr, i0 + chunk);
  int c[4] = {0, 0, 0, 0};
  for (int i = i0; i < i1; ++i) ++c[rowcls[i]];
  for (int q = 0; q < 4; ++q) cnt[t][q] = c[q];
  __syncthreads();
  if (t < 4) {  // exclusive prefix of class t over the threads
    int run = 0;
    for (int u = 0; u < 256; ++u) {
      const int v = cnt[u][t];
      cnt[u][t] = run;
      run += v;
    }
    cls_off[t] = run;  // (total of the class, for now)
  }
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int q = 0; q < 4; ++q) {
      const int v = cls_off[q];
      cls_off[q] = run;
      run += v;
    }
  }
  __syncthreads();
  int at[4];
  for (int q = 0; q < 4; ++q) at[q] = cls_off[q] + cnt[t][q];
  for (int i = i0; i < i1; ++i) {
    const int j = at[rowcls[i]]++;
    vpos[i] = j;
    vrow[j] = i;
  }
}

// key = ((cls * 8 + x) * NP + p) * R + g : the C blocks of class cls in row i = vrow[8 g + x] inside column panel p
__global__ void __launch_bounds__(256) order_count_cls(const uint32_t* __restrict__ c_bm, const unsigned char* __restrict__ rowcls,
                                                       const uint32_t* __restrict__ ncls_bm, const int* __restrict__ vrow, int nbr, int W, int PW, int NP, int R,
                                                       int* __restrict__ cnt) {
  const int key = blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= kNumClasses * 8 * NP * R) return;
  const int g = key % R, p = (key / R) % NP, x = (key / (R * NP)) % 8, cls = key / (R * NP * 8);
  const int j = 8 * g + x;
  int c = 0;
  if (j < nbr) {
    const int i = vrow[j];
    const int rc = rowcls[i];
    const int w1 = min(W, (p + 1) * PW);
    for (int w = p * PW; w < w1; ++w) c += __popc(c_bm[(size_t)i * W + w] & class_mask(cls, rc, ncls_bm, W, w));
  }
  cnt[key] = c;
}

// per class: common padded length of its eight XCD streams (multiple of 4) and the offset of its segment in order[]
__global__ void order_len_cls(const int64_t* __restrict__ base, int64_t total, int NP, int R, int64_t* __restrict__ lens /* 10 lens, 10 offsets, total */) {
  int64_t off = 0;
  for (int cls = 0; cls < kNumClasses; ++cls) {
    int64_t mx = 0;
    for (int x = 0; x < 8; ++x) {
      const size_t k0 = ((size_t)cls * 8 + x) * NP * R, k1 = k0 + (size_t)NP * R;
      const int64_t b1 = (cls == kNumClasses - 1 && x == 7) ? total : base[k1];
      mx = b1 - base[k0] > mx ? b1 - base[k0] : mx;
    }
    const int64_t len = (mx + 31) & ~(int64_t)31;  // multiple of 4 waves x up to 8 blocks per wave
    lens[cls] = len;
    lens[kNumClasses + cls] = off;
    off += 8 * len;
  }
  lens[2 * kNumClasses] = off;
}

// thread per (row i, bitmap word w): position of each C block inside the stream of its class and XCD
__global__ void __launch_bounds__(256) order_fill_cls(const uint32_t* __restrict__ c_bm, const int* __restrict__ c_pre,
                                                      const int* __restrict__ c_row_p, const unsigned char* __restrict__ rowcls,
                                                      const uint32_t* __restrict__ ncls_bm, const int64_t* __restrict__ base,
                                                      const int64_t* __restrict__ lens, const int* __restrict__ vpos, int nbr, int W, int PW, int NP,
                                                      int R, int* __restrict__ order) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (int64_t)nbr * W) return;
  const int i = (int)(tid / W), w = (int)(tid % W);
  uint32_t v = c_bm[tid];
  if (!v) return;
  const int j = vpos[i];  // (class_row_deal: the row's place among the rows of its class decides its XCD)
  const int x = j & 7, g = j >> 3, p = w / PW, rc = rowcls[i];
  // blocks of each column class that precede word w inside the panel (class 9 of a row whose own size is unranked: all of them)
  int before[4] = {0, 0, 0, 0};
  for (int ww = p * PW; ww < w; ++ww) {
    const uint32_t cw = c_bm[(size_t)i * W + ww];
    if (rc == 3)
      before[3] += __popc(cw);
    else
      for (int q = 0; q < 4; ++q) before[q] += __popc(cw & ncls_bm[(size_t)q * W + ww]);
  }
  int cb = c_row_p[i] + c_pre[tid];
  while (v) {
    const int bit = __ffs(v) - 1;
    v &= v - 1;
    int q = 3;
    if (rc != 3)
      for (int qq = 0; qq < 3; ++qq)
        if ((ncls_bm[(size_t)qq * W + w] >> bit) & 1u) q = qq;
    const int cls = (rc < 3 && q < 3) ? rc * 3 + q : 9;
    const size_t key = (((size_t)cls * 8 + x) * NP + p) * R + g;
    const size_t key0 = ((size_t)cls * 8 + x) * NP * R;
    const int64_t pos = lens[kNumClasses + cls] + (int64_t)x * lens[cls] + (base[key] - base[key0]) + before[q];
    order[pos] = cb;
    ++before[q];
    ++cb;
  }
}

}  // namespace dbcsr_amd
#endif
