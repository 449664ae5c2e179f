// mm_engine.hip -- device-resident local multiply (include/dbcsr_amd_mm.h).
//
// Symbolic phase (integer, HBM/L2-bound): block-level bitmaps.
//   Bbm[k][w]  : bit j set iff B(k,j) present                 (bitmap_from_index)
//   Cbm[i][w]  = Cin_bm[i][w] | OR_{k in A-row(i)} Bbm[k][w]  (c_bitmap)
//   row prefix popcounts give, without any hashing, the sorted column index of
//   C (what dbcsr_finalize produces, work/dbcsr_work_operations.F:749+) and the
//   rank of any block inside its row (row_prefix).
//   For every C block the list of products (a_off, b_off, k) is emitted in
//   ascending k: deterministic, no atomics (count_products / fill_products).
// This restates WHAT dbcsr_mm_csr_multiply_low computes (mm/dbcsr_mm_csr.F:
// 257-357: which C blocks exist, which (A,B) pairs feed each) with a data-
// parallel algorithm instead of its per-thread hash tables and 30000-entry
// parameter stacks.
//
// Numeric phase (fp64/fp32 MFMA): one wavefront per C block, all products of
// the block accumulated in registers, C written exactly once (no atomics, no
// zero-fill pass, bitwise reproducible).  Kernels, chosen per launch by the host code at the end of this file:
//   mm_numeric_f64_hot<M,N,K> / mm_numeric_f32_hot<M,N,K>  exact-size kernels, one (m, n, k) dominates (cubes 9..32)
//   mm_numeric_f64_tiny                                    C blocks of at most 4 x 4: four C blocks per wave
//   mm_numeric_f64_lds<MAXT> / mm_numeric_f64_pipe<MAXT>   any sizes up to 32 (pipe: mixed sizes, few products per block)
//   mm_numeric_f32_lds                                     fp32, any sizes up to 32
//   mm_numeric_f64 / mm_numeric_f32                        blocks above 32 (32 x 32 tiles, fragments from global memory)
// Around them: transpose, checksum, synthetic fill, norm filter, crop / window scale (submatrix limits).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <vector>

#include "../../include/dbcsr_amd_mm.h"
#include "common.h"
#include "smm_core.h"
#include "mm_types.h"
#include "mm_jit.h"

namespace dbcsr_amd {

// ----------------------------------------------------------------------------
// small utilities
// ----------------------------------------------------------------------------
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
    if (e != hipSuccess) return check(e, "hipMalloc(workspace)", __FILE__, __LINE__);
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};


// ----------------------------------------------------------------------------
// exclusive scan (int32 in -> TO out), three small kernels
// ----------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanChunk = kScanThreads * kScanItems;

__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t* total) {
  __shared__ int64_t wsum[kScanThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int64_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int64_t woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kScanThreads / 64; ++i) {
    if (i < w) woff += wsum[i];
    tot += wsum[i];
  }
  __syncthreads();
  *total = tot;
  return woff + inc - v;
}

__global__ void __launch_bounds__(kScanThreads) scan_reduce(const int* __restrict__ in, int64_t n, int64_t* __restrict__ partial) {
  const int64_t base = (int64_t)blockIdx.x * kScanChunk;
  int64_t s = 0;
  for (int it = 0; it < kScanItems; ++it) {
    const int64_t i = base + (int64_t)it * kScanThreads + threadIdx.x;
    if (i < n) s += in[i];
  }
  int64_t tot;
  (void)block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kScanThreads) scan_partials(int64_t* __restrict__ partial, int np, int64_t* __restrict__ total_out) {
  int64_t carry = 0;
  for (int base = 0; base < np; base += kScanThreads) {
    const int i = base + threadIdx.x;
    const int64_t v = i < np ? partial[i] : 0;
    int64_t tot;
    const int64_t ex = block_exclusive_scan(v, &tot);
    if (i < np) partial[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename TO>
__global__ void __launch_bounds__(kScanThreads) scan_apply(const int* __restrict__ in, int64_t n, const int64_t* __restrict__ partial,
                                                            TO* __restrict__ out, int write_total_at_n) {
  const int64_t base = (int64_t)blockIdx.x * kScanChunk;
  // thread t owns kScanItems consecutive items
  const int64_t first = base + (int64_t)threadIdx.x * kScanItems;
  int v[kScanItems];
  int64_t s = 0;
#pragma unroll
  for (int it = 0; it < kScanItems; ++it) {
    const int64_t i = first + it;
    v[it] = i < n ? in[i] : 0;
    s += v[it];
  }
  int64_t tot;
  int64_t ex = block_exclusive_scan(s, &tot) + partial[blockIdx.x];
#pragma unroll
  for (int it = 0; it < kScanItems; ++it) {
    const int64_t i = first + it;
    if (i < n) out[i] = (TO)ex;
    ex += v[it];
    if (write_total_at_n && i == n - 1) out[n] = (TO)ex;
  }
}

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
// numeric kernels
// ----------------------------------------------------------------------------
template <int MA, int NC>
__device__ __forceinline__ void cblock_f64(const Desc& d, const Entry* __restrict__ entries, const double* __restrict__ a_data,
                                           const double* __restrict__ b_data, double* __restrict__ c_out,
                                           const double* __restrict__ c_in, double alpha, double beta, const LaneMap& L, int row0,
                                           int col0) {
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
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int row = row0 + 8 * a + L.rowd, col = col0 + 8 * c + L.coll;
      if (row < m && col < n) {
        double v = alpha * acc[a][c];
        if (has_in) v += beta * Ci[row + (size_t)m * col];
        C[row + (size_t)m * col] = v;
      }
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
// 2 = production with the fragment reads kept as single ds_read_b64 (the compiler pairs them into ds_read2_b64 otherwise)
template <int M, int N, int K, int VAR>
__device__ __forceinline__ void cblock_f64_exact(const Desc& d, const Entry first, const Entry* __restrict__ entries, const double* __restrict__ a_data,
                                                 const double* __restrict__ b_data, double* __restrict__ c_out,
                                                 const double* __restrict__ c_in, double alpha, double beta, const LaneMap& L, int lane,
                                                 char* lds_a, char* lds_b, int dbg_rt, double* __restrict__ norm_out) {
  const int dbg = VAR == 1 ? dbg_rt : 0;
  constexpr int MA = (M + 7) / 8, NC = (N + 7) / 8, KS = (K + 3) / 4, K4 = 4 * KS;
  constexpr int CA = (M * K4 * 8 + 1023) / 1024, CB = (K * N * 8 + 1023) / 1024;
  double acc[MA][NC];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = 0.0;
  const Entry* e = entries + d.prod_start;
  const int cnt = d.prod_cnt;
  u32x4 ra[CA], rb[CB];
  const int voff = lane * 16;
  // fragment addresses: constant for the whole life of the wave
  const double* pa[MA];
  const double* pb[NC];
  const double* pbt[NC];  // last k step when K is not a multiple of 4: lanes past the end read element (0, col) (A's padding is zero)
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    int row = 8 * a + L.rowl;
    row = row < M ? row : M - 1;
    pa[a] = reinterpret_cast<const double*>(lds_a) + row + M * L.kq;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = 8 * c + L.coll;
    col = col < N ? col : N - 1;
    pb[c] = reinterpret_cast<const double*>(lds_b) + L.kq + K * col;
    const int kt = 4 * (KS - 1) + L.kq;
    pbt[c] = reinterpret_cast<const double*>(lds_b) + (kt < K ? kt : 0) + K * col;
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
      for (int c = 0; c < CA; ++c) *reinterpret_cast<u32x4*>(lds_a + c * 1024 + voff) = ra[c];
      DBCSR_AMD_LDS_ORDER();
#pragma unroll
      for (int c = 0; c < CB; ++c) *reinterpret_cast<u32x4*>(lds_b + c * 1024 + voff) = rb[c];
    }
    while (i1 < cnt && e1.ks() != K) {
      ++i1;
      e1 = e[i1 < cnt ? i1 : cnt - 1];
    }
    if (i1 < cnt) issue(e1.a_off(), e1.b_off());
    const Entry e2 = e[i1 + 1 < cnt ? i1 + 1 : cnt - 1];
    if (!(dbg & 2))
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      double av[MA], bv[NC];
#pragma unroll
      for (int a = 0; a < MA; ++a) {
        if constexpr (VAR == 2)
          av[a] = *(lds_vd*)(pa[a] + s * 4 * M);
        else
          av[a] = pa[a][s * 4 * M];
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if constexpr (VAR == 2)
          bv[c] = (s == KS - 1 && (K & 3)) ? *(lds_vd*)(pbt[c]) : *(lds_vd*)(pb[c] + 4 * s);
        else
          bv[c] = (s == KS - 1 && (K & 3)) ? pbt[c][0] : pb[c][4 * s];
      }
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
    }
    i0 = i1;
    e0 = e1;
    i1 = i1 + 1;
    e1 = e2;
  }
  for (int p = 0; p < cnt; ++p) {
    const Entry ep = e[p];
    if (ep.ks() != K) block_product_f64<MA, NC, false>(acc, a_data + ep.a_off(), b_data + ep.b_off(), M, N, ep.ks(), L);
  }
  const bool has_in = d.cin_off >= 0;
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
      if (norm_out) *reinterpret_cast<f64x2*>(lds_a + c * 1024 + voff) = v;  // (the final values, for the norm below)
      if (dbg & 16)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsc, voff, c * 1024, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsc, voff, c * 1024, 2);
    }
  } else {
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds_a + c * 1024 + voff);
      if (dbg & 16)
        __builtin_amdgcn_raw_buffer_store_b128(v, rsc, voff, c * 1024, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b128(v, rsc, voff, c * 1024, 2);
    }
  }
  // squared Frobenius norm of the block as it was stored (the final block filter of a filtered multiply reads it instead of C):
  // the block still sits in the wave's LDS slice
  if (norm_out) {
    double ss = 0.0;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const f64x2 v = *reinterpret_cast<const f64x2*>(lds_a + c * 1024 + voff);
      const int idx = c * 128 + 2 * lane;
      if (idx < M * N) ss += v[0] * v[0];
      if (idx + 1 < M * N) ss += v[1] * v[1];
    }
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
  const Desc d = {w.c_off, w.cin_off, w.prod_start, w.prod_cnt, w.m, w.n};
  Entry first;
  first.a_lo = w.a_lo, first.b_lo = w.b_lo, first.w = w.w;
  if ((dbg & 32) && d.prod_cnt == 0) return;
  if ((dbg & 64) && d.m == M && d.n == N) return;  // the tile kernel (mm_tile.h) computed the blocks of the dominant size
  char* lds_a = smem + (size_t)wid * lds_wave_doubles * 8;
  char* lds_b = lds_a + (size_t)lds_a_doubles * 8;
  const LaneMap L(lane);
  if (d.m == M && d.n == N) {
    cblock_f64_exact<M, N, K, VAR>(d, first, entries, a_data, b_data, c_out, c_in, alpha, beta, L, lane, lds_a, lds_b, dbg,
                              norms ? norms + w.cb : nullptr);
    return;
  }
  // the few blocks of another size (tail block row / column): straight from global memory, as one 32 x 32 tile
  cblock_f64<4, 4>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, L, 0, 0);
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

// max and (negated) min of an int array (block sizes): out[0] = max v, out[1] = max -v
__global__ void __launch_bounds__(256) max_of(const int* __restrict__ v, int n, int* __restrict__ out) {
  int mx = 0, mn = -0x7fffffff;
  for (int i = threadIdx.x; i < n; i += 256) {
    mx = max(mx, v[i]);
    mn = max(mn, -v[i]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mx = max(mx, __shfl_down(mx, off, 64));
    mn = max(mn, __shfl_down(mn, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out, mx);
    atomicMax(out + 1, mn);
  }
}

// most frequent value among the entries of v that lie in 1..32: out[0] = value (0: none), out[1] = how often
__global__ void __launch_bounds__(256) mode_of(const int* __restrict__ v, int n, int* __restrict__ out) {
  __shared__ int h[33];
  if (threadIdx.x < 33) h[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int s = v[i];
    if (s >= 1 && s <= 32) atomicAdd(&h[s], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int best = 0, cnt = 0;
    for (int s = 1; s <= 32; ++s)
      if (h[s] > cnt) {
        cnt = h[s];
        best = s;
      }
    out[0] = best;
    out[1] = cnt;
  }
}

// ---- fp64, C blocks of at most 4 x 4 (BASELINE config 1: 4 x 4 x 4 blocks) -------------------------------------------
// v_mfma_f64_4x4x4_4b_f64 multiplies FOUR independent 4x4x4 block triples at once (lane bits 2-3 select the triple).  With
// one wave per C block three quarters of every instruction are padding and a wave lives for ten products; here a wave
// owns four C blocks, one per MFMA sub-block, each walking its own product list, and every lane fetches exactly the A and
// B element it feeds (no LDS, no staging): per block product 2 element loads per lane and one MFMA per 4 of k.
// Sub-block b of lane l: b = (l >> 2) & 3; operands A[i = l & 3][k = l >> 4], B[k = l >> 4][j = l & 3]; result C[i = l >> 4][j = l & 3].
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
  double acc = 0.0;
  Entry cur = Entry::make(0, 0, 0);
  if (cnt > 0) cur = e[0];
  for (int p = 0; __any(p < cnt); ++p) {
    const bool on = p < cnt;
    Entry nxt = cur;
    if (p + 1 < cnt) nxt = e[p + 1];  // requested before this product's elements: one entry ahead
    const int ks = on ? cur.ks() : 0;
    const double* A = a_data + cur.a_off();
    const double* B = b_data + cur.b_off();
    for (int kb = 0; __any(kb < ks); kb += 4) {
      const int k = kb + kq;
      const bool kv = k < ks;
      const double av = (kv && x < m) ? A[x + m * k] : 0.0;
      const double bv = (kv && x < n) ? B[k + ks * x] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc, 0, 0, 0);
    }
    cur = nxt;
  }
  if (!live || (skip_empty && cnt == 0)) return;
  const int i = kq, j = x;
  if (i < m && j < n) {
    double v = alpha * acc;
    if (d.cin_off >= 0) v += beta * c_in[d.cin_off + i + m * j];
    c_out[d.c_off + i + m * j] = v;
  }
}

// ---- fp32, LDS-staged (blocks up to 32 x 32; BASELINE config 5) -----------------------------------------
// One v_mfma_f32_32x32x2_f32 covers the whole C block for 2 k.  A (m x k, column-major) is copied to LDS as
// is: its fragment (lane = row, two k per instruction) reads 32 consecutive floats.  B is stored k x n with k
// contiguous, but its fragment wants n across lanes at a fixed k -- 32 lanes 128 B apart would all hit one LDS
// bank -- so B is written to LDS TRANSPOSED with a row pitch of 33 floats (Bt[j + 33 kk]); the staging write
// computes (kk, j) per element with a multiply-shift division by the runtime k.
constexpr int F32_CH = 4;            // 1 KiB chunks: 4 x 256 floats >= 32 x 32
constexpr int F32_LDN = 33;          // pitch of the transposed B image
constexpr int F32_A_FLOATS = 1024 + 64, F32_BT_FLOATS = F32_LDN * 32 + 31;
constexpr int F32_WAVE_FLOATS = F32_A_FLOATS + ((F32_BT_FLOATS + 3) & ~3);

__device__ __forceinline__ void cblock_f32_lds(const Desc& d, const Entry* __restrict__ entries, const float* __restrict__ a_data,
                                               const float* __restrict__ b_data, float* __restrict__ c_out,
                                               const float* __restrict__ c_in, float alpha, float beta, int lane, float* lds_a,
                                               float* lds_bt) {
  constexpr int CH = F32_CH, LDN = F32_LDN;
  const int m = d.m, n = d.n, cnt = d.prod_cnt;
  const Entry* e = entries + d.prod_start;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  u32x4 ra[CH], rb[CH];
  const int voff = lane * 16;
  auto issue = [&](int p) {
    const int ks = e[p].ks();
    const int abytes = __builtin_amdgcn_readfirstlane(m * ks * 4), bbytes = __builtin_amdgcn_readfirstlane(ks * n * 4);  // scalar on purpose, see cblock_f64_lds
    const int nca = __builtin_amdgcn_readfirstlane((m * (ks + 1) * 4 + 1023) >> 10), ncb = (bbytes + 1023) >> 10;  // A: one zero column of k padding
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + e[p].a_off()), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + e[p].b_off()), 0, bbytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < nca) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < ncb) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
  };
  if (cnt > 0) issue(0);
  const int i = lane & 31, kh = lane >> 5;
  const int arow = i < m ? i : m - 1, bcol = i < n ? i : n - 1;
  for (int p = 0; p < cnt; ++p) {
    const int ks = e[p].ks();
    const int kn = ks * n;
    const int nca = __builtin_amdgcn_readfirstlane((m * (ks + 1) * 4 + 1023) >> 10), ncb = __builtin_amdgcn_readfirstlane((kn * 4 + 1023) >> 10);
    const unsigned inv = (65536u + (unsigned)ks - 1u) / (unsigned)ks;  // j = (e * inv) >> 16 == e / ks for e < 2048, ks <= 32
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < nca) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(lds_a) + c * 1024 + voff) = ra[c];
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < ncb) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const unsigned el = (unsigned)((c * 64 + lane) * 4 + t);
          const unsigned j = (el * inv) >> 16, kk = el - j * (unsigned)ks;
          if ((int)el < kn) lds_bt[j + LDN * kk] = __uint_as_float(rb[c][t]);
        }
      }
    if (p + 1 < cnt) issue(p + 1);
    // multiply: lane (i, kh) feeds A[i][2s + kh] and B[2s + kh][i]; the odd-k tail reads A's zero padding column
    const int nsteps = (ks + 1) >> 1;
    int aoff = arow + m * kh;
    for (int s2 = 0; s2 < nsteps; ++s2) {
      const int kk = 2 * s2 + kh;
      const float av = lds_a[aoff];
      const float bv = lds_bt[bcol + LDN * (kk < ks ? kk : ks - 1)];
      aoff += 2 * m;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  float* C = c_out + d.c_off;
  const bool has_in = d.cin_off >= 0;
  const float* Ci = c_in + (has_in ? d.cin_off : 0);
  const int col = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < m && col < n) {
      float v = alpha * acc[r];
      if (has_in) v += beta * Ci[row + (size_t)m * col];
      C[row + (size_t)m * col] = v;
    }
  }
}

// Exact-size fp32 variant (see cblock_f64_exact): with M, N, K known at compile time the transposed LDS image of B needs no
// per-element division (the generic kernel spends 208 VALU + 140 SALU instructions per 32^3 product next to 16 MFMAs,
// MFMA pipe 42 % busy): the LDS address of every staged element is a per-wave constant.
template <int M, int N, int K>
__device__ __forceinline__ void cblock_f32_exact(const Desc& d, const Entry* __restrict__ entries, const float* __restrict__ a_data,
                                                 const float* __restrict__ b_data, float* __restrict__ c_out,
                                                 const float* __restrict__ c_in, float alpha, float beta, int lane, float* lds_a,
                                                 float* lds_bt) {
  constexpr int LDN = F32_LDN;
  constexpr int KS2 = (K + 1) / 2, KP = 2 * KS2;                       // k steps of 2; A is zero-padded to KP columns
  constexpr int CA = (M * KP * 4 + 1023) / 1024, CB = (K * N * 4 + 1023) / 1024;
  constexpr int DUMMY = F32_BT_FLOATS;                                 // LDS slot that swallows the staging lanes past the block end
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const Entry* e = entries + d.prod_start;
  const int cnt = d.prod_cnt;
  u32x4 ra[CA], rb[CB];
  const int voff = lane * 16;
  int baddr[CB][4];  // where element t of chunk c of this lane goes in the transposed image: constants of the wave
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int el = (c * 64 + lane) * 4 + t;
      const int j = el / K, kk = el - j * K;
      baddr[c][t] = el < K * N ? j + LDN * kk : DUMMY;
    }
  const int i = lane & 31, kh = lane >> 5;
  const float* pa = lds_a + (i < M ? i : M - 1) + M * kh;
  const float* pb = lds_bt + (i < N ? i : N - 1) + LDN * kh;
  const float* pbt = lds_bt + (i < N ? i : N - 1) + LDN * ((K & 1) && kh ? K - 1 : 2 * (KS2 - 1) + kh);  // last step of an odd K
  auto issue = [&](uint64_t a_off, uint64_t b_off) {
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + a_off), 0, M * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + b_off), 0, K * N * 4, 0x00020000);
#pragma unroll
    for (int c = 0; c < CA; ++c) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CB; ++c) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
  };
  int i0 = 0;
  Entry e0 = e[0];
  while (i0 < cnt && e0.ks() != K) {
    ++i0;
    e0 = e[i0 < cnt ? i0 : cnt - 1];
  }
  int i1 = i0 + 1;
  Entry e1 = e[i1 < cnt ? i1 : cnt - 1];
  if (i0 < cnt) issue(e0.a_off(), e0.b_off());
  while (i0 < cnt) {
#pragma unroll
    for (int c = 0; c < CA; ++c) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(lds_a) + c * 1024 + voff) = ra[c];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) lds_bt[baddr[c][t]] = __uint_as_float(rb[c][t]);
    while (i1 < cnt && e1.ks() != K) {
      ++i1;
      e1 = e[i1 < cnt ? i1 : cnt - 1];
    }
    if (i1 < cnt) issue(e1.a_off(), e1.b_off());
    const Entry e2 = e[i1 + 1 < cnt ? i1 + 1 : cnt - 1];
#pragma unroll
    for (int s2 = 0; s2 < KS2; ++s2) {
      const float av = pa[s2 * 2 * M];
      const float bv = (s2 == KS2 - 1) ? pbt[0] : pb[s2 * 2 * LDN];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    i0 = i1;
    e0 = e1;
    i1 = i1 + 1;
    e1 = e2;
  }
  for (int p = 0; p < cnt; ++p) {
    const Entry ep = e[p];
    if (ep.ks() != K) block_product_f32<false>(acc, a_data + ep.a_off(), b_data + ep.b_off(), M, N, ep.ks(), lane);
  }
  float* C = c_out + d.c_off;
  const bool has_in = d.cin_off >= 0;
  const float* Ci = c_in + (has_in ? d.cin_off : 0);
  const int col = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < M && col < N) {
      float v = alpha * acc[r];
      if (has_in) v += beta * Ci[row + (size_t)M * col];
      C[row + (size_t)M * col] = v;
    }
  }
}

static inline size_t f32_lds_bytes(int wg_waves) { return ((size_t)wg_waves * F32_WAVE_FLOATS + 4) * sizeof(float); }
#define DBCSR_F32_KERNEL_HEAD                                                                          \
  extern __shared__ __attribute__((aligned(16))) char smem_raw_[]; /* f32_lds_bytes(waves per workgroup) */ \
  float* smem = reinterpret_cast<float*>(smem_raw_);                                                   \
  const int lane = threadIdx.x & 63;                                                                   \
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                             \
  const int wg = xcd_remap(blockIdx.x, gridDim.x);                                                     \
  const int64_t pos = (int64_t)wg * (int)(blockDim.x >> 6) + wid;                                      \
  const int64_t cb = order[pos];                                                                       \
  if (cb < 0 || cb >= nblk) return;                                                                    \
  const Desc d = descs[cb];                                                                            \
  if (skip_empty && d.prod_cnt == 0) return;                                                           \
  float* lds_a = smem + (size_t)wid * F32_WAVE_FLOATS;                                                 \
  float* lds_bt = lds_a + F32_A_FLOATS;

__global__ void __launch_bounds__(256) mm_numeric_f32_lds(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                          const float* __restrict__ a_data, const float* __restrict__ b_data,
                                                          float* __restrict__ c_out, const float* __restrict__ c_in, float alpha,
                                                          float beta, int skip_empty, const int* __restrict__ order) {
  DBCSR_F32_KERNEL_HEAD
  cblock_f32_lds(d, entries, a_data, b_data, c_out, c_in, alpha, beta, lane, lds_a, lds_bt);
}

template <int M, int N, int K>
__global__ void __launch_bounds__(256) mm_numeric_f32_hot(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                          const float* __restrict__ a_data, const float* __restrict__ b_data,
                                                          float* __restrict__ c_out, const float* __restrict__ c_in, float alpha,
                                                          float beta, int skip_empty, const int* __restrict__ order) {
  DBCSR_F32_KERNEL_HEAD
  if (d.m == M && d.n == N)
    cblock_f32_exact<M, N, K>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, lane, lds_a, lds_bt);
  else
    cblock_f32_lds(d, entries, a_data, b_data, c_out, c_in, alpha, beta, lane, lds_a, lds_bt);
}

__global__ void __launch_bounds__(256) mm_numeric_f32(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                      const float* __restrict__ a_data, const float* __restrict__ b_data,
                                                      float* __restrict__ c_out, const float* __restrict__ c_in, float alpha,
                                                      float beta, int skip_empty) {
  const int lane = threadIdx.x & 63;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t cb = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6));
  if (cb >= nblk) return;
  const Desc d = descs[cb];
  if (skip_empty && d.prod_cnt == 0) return;
  const int m = d.m, n = d.n;
  const Entry* e = entries + d.prod_start;
  const bool has_in = d.cin_off >= 0;
  for (int row0 = 0; row0 < m; row0 += 32)
    for (int col0 = 0; col0 < n; col0 += 32) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      for (int p = 0; p < d.prod_cnt; ++p)
        block_product_f32<false>(acc, a_data + e[p].a_off(), b_data + e[p].b_off(), m, n, e[p].ks(), lane, row0, col0);
      float* C = c_out + d.c_off;
      const float* Ci = c_in + (has_in ? d.cin_off : 0);
      const int col = col0 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < m && col < n) {
          float v = alpha * acc[r];
          if (has_in) v += beta * Ci[row + (size_t)m * col];
          C[row + (size_t)m * col] = v;
        }
      }
    }
}

// ----------------------------------------------------------------------------
// auxiliary kernels: checksum, random fill, transpose
// ----------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) checksum_blocks(const int* __restrict__ row_p, const int* __restrict__ col_i,
                                                       const int64_t* __restrict__ blk_p, const T* __restrict__ data,
                                                       const int* __restrict__ rs, const int* __restrict__ cs,
                                                       const int64_t* __restrict__ roff, const int64_t* __restrict__ coff, int nbr,
                                                       double* __restrict__ row_sums) {
  // one wavefront per block row; fixed summation order -> reproducible
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= nbr) return;
  const int m = rs[row];
  double s2 = 0.0, sp = 0.0;
  for (int b = row_p[row]; b < row_p[row + 1]; ++b) {
    const int c = col_i[b];
    const int n = cs[c];
    const T* d = data + blk_p[b];
    for (int e = lane; e < m * n; e += 64) {
      const double x = (double)d[e];
      const int r = e % m, cc = e / m;
      s2 += x * x;
      sp += x * log(fabs((double)(roff[row] + r + 1) * (double)(coff[c] + cc + 1)));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s2 += __shfl_down(s2, off, 64);
    sp += __shfl_down(sp, off, 64);
  }
  if (lane == 0) {
    row_sums[2 * row] = s2;
    row_sums[2 * row + 1] = sp;
  }
}

__global__ void __launch_bounds__(256) checksum_final(const double* __restrict__ row_sums, int nbr, double* __restrict__ out2) {
  __shared__ double r2[256], rp[256];
  double s2 = 0.0, sp = 0.0;
  for (int i = threadIdx.x; i < nbr; i += 256) {
    s2 += row_sums[2 * i];
    sp += row_sums[2 * i + 1];
  }
  r2[threadIdx.x] = s2;
  rp[threadIdx.x] = sp;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      r2[threadIdx.x] += r2[threadIdx.x + off];
      rp[threadIdx.x] += rp[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = r2[0];
    out2[1] = rp[0];
  }
}

// LAPACK xLARUV stream: x_i = seed * a^i mod 2^48 (a = 33952834046453); see
// oracle/dbcsr_oracle.c for the statement of the published algorithm.
__device__ __forceinline__ uint64_t larnv_block_seed(int irow, int nrow, int icol, int ival) {
  // set_larnv_seed, src/utils/dbcsr_blas_operations.F:29-52 (irow/icol 1-based)
  long long ivm = ((long long)ival) % 65536;
  if (ivm < 0) ivm += 65536;
  long long map = (((long long)irow - 1 + (long long)icol * (long long)nrow) * (1 + ivm)) * 2 + 1;
  const uint64_t s4 = (uint64_t)(map % 4096);
  map /= 4096;
  const uint64_t s3 = (uint64_t)((map ^ 3541) % 4096);
  map /= 4096;
  const uint64_t s2 = (uint64_t)((map ^ 1153) % 4096);
  map /= 4096;
  const uint64_t s1 = (uint64_t)((map ^ 2029) % 4096);
  return (s1 << 36) | (s2 << 24) | (s3 << 12) | s4;
}

__device__ __forceinline__ uint64_t pow48(uint64_t base, uint64_t e) {
  const uint64_t mask = (1ull << 48) - 1;
  uint64_t r = 1;
  base &= mask;
  while (e) {
    if (e & 1) r = (r * base) & mask;
    base = (base * base) & mask;
    e >>= 1;
  }
  return r;
}

__global__ void __launch_bounds__(256) fill_random_f64(const int* __restrict__ row_p, const int* __restrict__ col_i,
                                                       const int64_t* __restrict__ blk_p, double* __restrict__ data,
                                                       const int* __restrict__ rs, const int* __restrict__ cs, int nbr, int nbc,
                                                       int counter, const int* __restrict__ row_gid, const int* __restrict__ col_gid,
                                                       int nrow_global) {
  // one wavefront per block row, lanes over the elements of each block
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= nbr) return;
  (void)nbc;
  const uint64_t mask = (1ull << 48) - 1, A = 33952834046453ull;
  const uint64_t a64 = pow48(A, 64);
  for (int b = row_p[row]; b < row_p[row + 1]; ++b) {
    const int c = col_i[b];
    const int ne = rs[row] * cs[c];
    const uint64_t seed = larnv_block_seed((row_gid ? row_gid[row] : row) + 1, nrow_global, (col_gid ? col_gid[c] : c) + 1, counter);
    uint64_t x = (seed * pow48(A, (uint64_t)lane + 1)) & mask;
    double* d = data + blk_p[b];
    for (int e = lane; e < ne; e += 64) {
      d[e] = (double)x * (1.0 / 281474976710656.0);
      x = (x * a64) & mask;
    }
  }
}

__global__ void __launch_bounds__(256) fill_random_f32(const int* __restrict__ row_p, const int* __restrict__ col_i,
                                                       const int64_t* __restrict__ blk_p, float* __restrict__ data,
                                                       const int* __restrict__ rs, const int* __restrict__ cs, int nbr, int nbc,
                                                       int counter, const int* __restrict__ row_gid, const int* __restrict__ col_gid,
                                                       int nrow_global) {
  // slarnv draws in chunks of 64 and, inside a chunk, a value that rounds to 1.0f
  // bumps the chunk's base seed (LAPACK slaruv) -- so one thread walks one block.
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  (void)nbc;
  // thread per block: find its row by binary search in row_p
  const int64_t nblks = row_p[nbr];
  if (t >= nblks) return;
  int lo = 0, hi = nbr;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (row_p[mid] <= t) lo = mid; else hi = mid;
  }
  const int r = lo, c = col_i[t];
  const int ne = rs[r] * cs[c];
  const uint64_t mask = (1ull << 48) - 1, A = 33952834046453ull;
  uint64_t seed = larnv_block_seed((row_gid ? row_gid[r] : r) + 1, nrow_global, (col_gid ? col_gid[c] : c) + 1, counter);
  float* d = data + blk_p[t];
  const float rr = 1.0f / 4096.0f;
  for (int done = 0; done < ne; done += 64) {
    const int il = (ne - done) < 64 ? (ne - done) : 64;
    // limbs of the chunk's base seed (may exceed 4095 after a bump)
    long long i1 = (long long)((seed >> 36) & 4095), i2 = (long long)((seed >> 24) & 4095), i3 = (long long)((seed >> 12) & 4095),
              i4 = (long long)(seed & 4095);
    uint64_t apow = 1, last = 0;
    for (int i = 0; i < il; ++i) {
      apow = (apow * A) & mask;
      for (;;) {
        const uint64_t full = ((uint64_t)i1 << 36) + ((uint64_t)i2 << 24) + ((uint64_t)i3 << 12) + (uint64_t)i4;
        const uint64_t p = (full * apow) & mask;
        const float v = rr * ((float)((p >> 36) & 4095) + rr * ((float)((p >> 24) & 4095) + rr * ((float)((p >> 12) & 4095) + rr * (float)(p & 4095))));
        if (v == 1.0f) {
          i1 += 2; i2 += 2; i3 += 2; i4 += 2;
          continue;
        }
        d[done + i] = v;
        last = p;
        break;
      }
    }
    seed = last;
  }
}

// transpose: dst block (c, r) <- src block (r, c)^T
template <typename T>
__global__ void __launch_bounds__(256)
transpose_fill(const int* __restrict__ s_row_p, const int* __restrict__ s_col_i, const int64_t* __restrict__ s_blk_p,
               const T* __restrict__ s_data, const int* __restrict__ s_rs, const int* __restrict__ s_cs, const uint32_t* __restrict__ t_bm,
               const int* __restrict__ t_pre, const int* __restrict__ t_row_p, const int64_t* __restrict__ t_blk_p_ws, int s_nbr, int Wt,
               int* __restrict__ t_col_i, int64_t* __restrict__ t_blk_p, T* __restrict__ t_data) {
  // one wavefront per source block row
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= s_nbr) return;
  const int m = s_rs[r];
  for (int b = s_row_p[r]; b < s_row_p[r + 1]; ++b) {
    const int c = s_col_i[b];
    const int n = s_cs[c];
    // position of (c, r) in the transposed index
    const uint32_t wv = t_bm[(size_t)c * Wt + (r >> 5)];
    const int tb = t_row_p[c] + t_pre[(size_t)c * Wt + (r >> 5)] + __popc(wv & ((1u << (r & 31)) - 1u));
    const int64_t toff = t_blk_p_ws[tb];
    if (lane == 0) {
      t_col_i[tb] = r;
      t_blk_p[tb] = toff;
    }
    const T* src = s_data + s_blk_p[b];
    T* dst = t_data + toff;
    for (int e = lane; e < m * n; e += 64) {
      const int i = e % m, j = e / m;  // src(i, j) -> dst(j, i), dst is n x m
      dst[j + (size_t)n * i] = src[e];
    }
  }
}

__global__ void __launch_bounds__(256) transpose_mark(const int* __restrict__ s_row_p, const int* __restrict__ s_col_i, int s_nbr, int Wt,
                                                      uint32_t* __restrict__ t_bm) {
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= s_nbr) return;
  for (int b = s_row_p[r] + lane; b < s_row_p[r + 1]; b += 64) atomicOr(&t_bm[(size_t)s_col_i[b] * Wt + (r >> 5)], 1u << (r & 31));
}

// thread per (row c of the transposed matrix, word w): block sizes in index order
__global__ void __launch_bounds__(256) transpose_sizes(const uint32_t* __restrict__ t_bm, const int* __restrict__ t_pre,
                                                       const int* __restrict__ t_row_p, const int* __restrict__ s_rs,
                                                       const int* __restrict__ s_cs, int t_nbr, int Wt, int* __restrict__ blk_nze) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)t_nbr * Wt) return;
  const int c = (int)(t / Wt), w = (int)(t % Wt);
  uint32_t v = t_bm[t];
  int tb = t_row_p[c] + t_pre[t];
  while (v) {
    const int bit = __ffs(v) - 1;
    v &= v - 1;
    blk_nze[tb++] = s_cs[c] * s_rs[32 * w + bit];
  }
}


// ---- desymmetrize (dbcsr_desymmetrize_deep, what make_images does to a symmetric operand: src/mm/dbcsr_mm_cannon.F:284,
// 351-379): a symmetric / antisymmetric matrix stores one triangle; the full matrix has block (c, r) = +-block (r, c)^T too
// mode 0: desymmetrize (a block and its twin); mode 1: stored triangle -> canonical (checkerboard) form of a matrix with symmetry
// (dbcsr_make_index_canonical: block (r, c), r != c, moves to (c, r) when checker_tr says its twin is the stored one,
// src/dist/dbcsr_dist_operations.F:65-75 on the 1-based coordinates); mode 2: canonical form -> stored triangle (row <= col)
__device__ __forceinline__ bool twin_moves(int mode, int r, int c) {
  if (mode == 1) return r != c && ((((r + c) & 1) == 1) == (c >= r));
  return r > c;  // mode 2
}

__global__ void __launch_bounds__(256) desym_mark(const int* __restrict__ s_row_p, const int* __restrict__ s_col_i, int nbr, int W, int mode,
                                                  uint32_t* __restrict__ bm) {
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= nbr) return;
  for (int b = s_row_p[r] + lane; b < s_row_p[r + 1]; b += 64) {
    const int c = s_col_i[b];
    const bool stay = mode == 0 || !twin_moves(mode, r, c), go = mode == 0 || twin_moves(mode, r, c);
    if (stay) atomicOr(&bm[(size_t)r * W + (c >> 5)], 1u << (c & 31));
    if (go) atomicOr(&bm[(size_t)c * W + (r >> 5)], 1u << (r & 31));
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
desym_fill(const int* __restrict__ s_row_p, const int* __restrict__ s_col_i, const int64_t* __restrict__ s_blk_p, const T* __restrict__ s_data,
           const int* __restrict__ sizes, const uint32_t* __restrict__ bm, const int* __restrict__ pre, const int* __restrict__ d_row_p,
           const int64_t* __restrict__ d_blk_p_ws, int nbr, int W, T sign, int mode, int* __restrict__ d_col_i, int64_t* __restrict__ d_blk_p,
           T* __restrict__ d_data) {
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= nbr) return;
  const int m = sizes[r];
  auto slot = [&](int row, int col) {
    const uint32_t wv = bm[(size_t)row * W + (col >> 5)];
    return d_row_p[row] + pre[(size_t)row * W + (col >> 5)] + __popc(wv & ((1u << (col & 31)) - 1u));
  };
  for (int b = s_row_p[r]; b < s_row_p[r + 1]; ++b) {
    const int c = s_col_i[b];
    const int n = sizes[c];
    const T* src = s_data + s_blk_p[b];
    const bool stay = mode == 0 || !twin_moves(mode, r, c), go = mode == 0 ? c != r : twin_moves(mode, r, c);
    if (stay) {
      const int t0 = slot(r, c);
      if (lane == 0) {
        d_col_i[t0] = c;
        d_blk_p[t0] = d_blk_p_ws[t0];
      }
      T* d0 = d_data + d_blk_p_ws[t0];
      for (int e = lane; e < m * n; e += 64) d0[e] = src[e];
    }
    if (go) {
      const int t1 = slot(c, r);
      if (lane == 0) {
        d_col_i[t1] = r;
        d_blk_p[t1] = d_blk_p_ws[t1];
      }
      T* d1 = d_data + d_blk_p_ws[t1];
      for (int e = lane; e < m * n; e += 64) {
        const int i = e % m, j = e / m;  // src(i, j) -> dst(j, i), dst is n x m
        d1[j + (size_t)n * i] = sign * src[e];
      }
    }
  }
}

// ---- block filter (dbcsr_mm_multrec.F:694-748 multrec_filtering / dbcsr_filter): drop blocks with ||blk||^2 < eps^2
__global__ void __launch_bounds__(256) filter_flags(const double* __restrict__ norms64, int64_t nblks, const int* __restrict__ row_p,
                                                    const int* __restrict__ col_i, const int* __restrict__ rs, const int* __restrict__ cs,
                                                    int nbr, double eps2, int* __restrict__ keep, int* __restrict__ blk_nze,
                                                    int* __restrict__ row_keep) {
  // one wavefront per block row
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= nbr) return;
  int cnt = 0;
  for (int b = row_p[row] + lane; b < row_p[row + 1]; b += 64) {
    const int k = norms64[b] >= eps2 ? 1 : 0;
    keep[b] = k;
    blk_nze[b] = k ? rs[row] * cs[col_i[b]] : 0;
    cnt += k;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if (lane == 0) row_keep[row] = cnt;
  (void)nblks;
}

template <typename T>
__global__ void __launch_bounds__(256) filter_compact(const int* __restrict__ row_p, const int* __restrict__ col_i,
                                                      const int64_t* __restrict__ blk_p, const T* __restrict__ data,
                                                      const int* __restrict__ rs, const int* __restrict__ cs, int nbr, int S,
                                                      const int* __restrict__ keep, const int64_t* __restrict__ newidx,
                                                      const int64_t* __restrict__ newoff, int* __restrict__ d_col_i,
                                                      int64_t* __restrict__ d_blk_p, T* __restrict__ d_data) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int row = (int)(wv / S), sub = (int)(wv % S);
  if (row >= nbr) return;
  const int m = rs[row];
  for (int b = row_p[row] + sub; b < row_p[row + 1]; b += S) {
    if (!keep[b]) continue;
    const int64_t t = newidx[b], off = newoff[b];
    if (lane == 0) {
      d_col_i[t] = col_i[b];
      d_blk_p[t] = off;
    }
    const int ne = m * cs[col_i[b]];
    const T* src = data + blk_p[b];
    T* dst = d_data + off;
    for (int e = lane; e < ne; e += 64) dst[e] = src[e];
  }
}

// ---- submatrix limits (dbcsr_crop_matrix, src/ops/dbcsr_operations.F:1652-1833; dbcsr_scale with limits) ---------
struct Window {
  int r0, r1, c0, c1;  // inclusive 0-based element bounds
};

// one wavefront per block row: a block is kept when it intersects the window
__global__ void __launch_bounds__(256) crop_flags(const int* __restrict__ row_p, const int* __restrict__ col_i, const int* __restrict__ rs,
                                                  const int* __restrict__ cs, const int64_t* __restrict__ roff,
                                                  const int64_t* __restrict__ coff, int nbr, Window w, int* __restrict__ keep,
                                                  int* __restrict__ blk_nze, int* __restrict__ row_keep) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= nbr) return;
  const int m = rs[row];
  const bool row_in = roff[row] + m - 1 >= w.r0 && roff[row] <= w.r1;
  int cnt = 0;
  for (int b = row_p[row] + lane; b < row_p[row + 1]; b += 64) {
    const int c = col_i[b], n = cs[c];
    const int k = (row_in && coff[c] + n - 1 >= w.c0 && coff[c] <= w.c1) ? 1 : 0;
    keep[b] = k;
    blk_nze[b] = k ? m * n : 0;
    cnt += k;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if (lane == 0) row_keep[row] = cnt;
}

// compaction of the kept blocks; elements outside the window become zero
template <typename T>
__global__ void __launch_bounds__(256) crop_compact(const int* __restrict__ row_p, const int* __restrict__ col_i,
                                                    const int64_t* __restrict__ blk_p, const T* __restrict__ data,
                                                    const int* __restrict__ rs, const int* __restrict__ cs,
                                                    const int64_t* __restrict__ roff, const int64_t* __restrict__ coff, int nbr, Window w,
                                                    const int* __restrict__ keep, const int64_t* __restrict__ newidx,
                                                    const int64_t* __restrict__ newoff, int* __restrict__ d_col_i,
                                                    int64_t* __restrict__ d_blk_p, T* __restrict__ d_data) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= nbr) return;
  const int m = rs[row];
  const int64_t r_base = roff[row];
  for (int b = row_p[row]; b < row_p[row + 1]; ++b) {
    if (!keep[b]) continue;
    const int64_t t = newidx[b], off = newoff[b];
    const int c = col_i[b];
    if (lane == 0) {
      d_col_i[t] = c;
      d_blk_p[t] = off;
    }
    const int ne = m * cs[c];
    const int64_t c_base = coff[c];
    const T* src = data + blk_p[b];
    T* dst = d_data + off;
    for (int e = lane; e < ne; e += 64) {
      const int64_t gr = r_base + e % m, gc = c_base + e / m;
      dst[e] = (gr >= w.r0 && gr <= w.r1 && gc >= w.c0 && gc <= w.c1) ? src[e] : T(0);
    }
  }
}

// in place: x *= beta for the elements inside the window
template <typename T>
__global__ void __launch_bounds__(256) scale_window(const int* __restrict__ row_p, const int* __restrict__ col_i,
                                                    const int64_t* __restrict__ blk_p, T* __restrict__ data, const int* __restrict__ rs,
                                                    const int* __restrict__ cs, const int64_t* __restrict__ roff,
                                                    const int64_t* __restrict__ coff, int nbr, Window w, T beta) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= nbr) return;
  const int m = rs[row];
  const int64_t r_base = roff[row];
  if (r_base + m - 1 < w.r0 || r_base > w.r1) return;
  for (int b = row_p[row]; b < row_p[row + 1]; ++b) {
    const int c = col_i[b], n = cs[c];
    const int64_t c_base = coff[c];
    if (c_base + n - 1 < w.c0 || c_base > w.c1) continue;
    T* blk = data + blk_p[b];
    for (int e = lane; e < m * n; e += 64) {
      const int64_t gr = r_base + e % m, gc = c_base + e / m;
      if (gr >= w.r0 && gr <= w.r1 && gc >= w.c0 && gc <= w.c1) blk[e] *= beta;
    }
  }
}


// ----------------------------------------------------------------------------
// (m, n) classes of C blocks (mixed block sizes): order[] in one segment per class
//
// The reference sorts block products into homogeneous stacks by the three most common sizes of each dimension
// (map_most_common, src/dist/dbcsr_dist_util.F:753-812; stack_map in dbcsr_mm_csr.F:497-525) and runs each stack on the
// kernel compiled for its (m, n, k).  Here a C block is the unit of work, so C blocks are bucketed by (m, n): class
// c = 3 * rank(m) + rank(n) for the three most common row and column block sizes (ranks 0..2), class 9 = everything else.
// order[] becomes ten segments, each laid out like the single list of the other kernels (eight XCD streams padded to a
// common length, column panels, row i on XCD i mod 8), and each segment is one launch of the kernel for its class.
// ----------------------------------------------------------------------------
constexpr int kNumClasses = 10;

__global__ void __launch_bounds__(256) size_hist(const int* __restrict__ sizes, int n, int* __restrict__ hist /* 33 */) {
  // per-workgroup histogram in LDS first: with a single block size every global atomic would hit the same address
  // (5699 serialised atomics = 66 us on config 4)
  __shared__ int h[33];
  if (threadIdx.x < 33) h[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int s = sizes[i];
    atomicAdd(&h[(s >= 1 && s <= 32) ? s : 0], 1);
  }
  __syncthreads();
  if (threadIdx.x < 33 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

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

// key = ((cls * 8 + x) * NP + p) * R + g : the C blocks of class cls in row i = 8 g + x inside column panel p
__global__ void __launch_bounds__(256) order_count_cls(const uint32_t* __restrict__ c_bm, const unsigned char* __restrict__ rowcls,
                                                       const uint32_t* __restrict__ ncls_bm, int nbr, int W, int PW, int NP, int R,
                                                       int* __restrict__ cnt) {
  const int key = blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= kNumClasses * 8 * NP * R) return;
  const int g = key % R, p = (key / R) % NP, x = (key / (R * NP)) % 8, cls = key / (R * NP * 8);
  const int i = 8 * g + x;
  int c = 0;
  if (i < nbr) {
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
                                                      const int64_t* __restrict__ lens, int nbr, int W, int PW, int NP, int R,
                                                      int* __restrict__ order) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (int64_t)nbr * W) return;
  const int i = (int)(tid / W), w = (int)(tid % W);
  uint32_t v = c_bm[tid];
  if (!v) return;
  const int x = i & 7, g = i >> 3, p = w / PW, rc = rowcls[i];
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

// ---- per-(m, n, k) statistics (dbcsr_mm_sched.F:392-461): histogram over the product lists, open addressing ------------
constexpr int kStatSlots = 8192;  // power of two
// launch-order work records (mm_types.h Work): one thread per position of order[]
__global__ void __launch_bounds__(256) build_work(const int* __restrict__ order, int64_t npos, const Desc* __restrict__ descs, int64_t nblk,
                                                  const Entry* __restrict__ entries, Work* __restrict__ work) {
  const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= npos) return;
  const int cb = order[pos];
  Work w;
  w.c_off = 0, w.cin_off = -1, w.prod_start = 0, w.prod_cnt = -1, w.m = 0, w.n = 0, w.a_lo = 0, w.b_lo = 0, w.w = 0, w.cb = cb;
  if (cb >= 0 && cb < nblk) {
    const Desc d = descs[cb];
    w.c_off = d.c_off, w.cin_off = d.cin_off, w.prod_start = d.prod_start, w.prod_cnt = d.prod_cnt, w.m = d.m, w.n = d.n;
    if (d.prod_cnt > 0) {
      const Entry e = entries[d.prod_start];
      w.a_lo = e.a_lo, w.b_lo = e.b_lo, w.w = e.w;
    }
  }
  work[pos] = w;
}

__global__ void __launch_bounds__(256) mnk_histogram(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                     unsigned long long* __restrict__ keys, unsigned long long* __restrict__ counts,
                                                     int* __restrict__ overflow) {
  const int64_t cb = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cb >= nblk) return;
  const Desc d = descs[cb];
  const Entry* e = entries + d.prod_start;
  unsigned long long run_key = 0, run_cnt = 0;
  auto flush = [&]() {
    if (!run_cnt) return;
    unsigned h = (unsigned)((run_key * 0x9E3779B97F4A7C15ull) >> 40) & (kStatSlots - 1);
    for (int probe = 0; probe < kStatSlots; ++probe) {
      const unsigned long long prev = atomicCAS(&keys[h], 0ull, run_key);
      if (prev == 0ull || prev == run_key) {
        atomicAdd(&counts[h], run_cnt);
        return;
      }
      h = (h + 1) & (kStatSlots - 1);
    }
    *overflow = 1;
  };
  for (int p = 0; p < d.prod_cnt; ++p) {
    // key: m | n << 16 | k << 32, plus bit 63 so that no key is 0
    const unsigned long long key = (unsigned long long)(uint16_t)d.m | ((unsigned long long)(uint16_t)d.n << 16) |
                                   ((unsigned long long)(unsigned)e[p].ks() << 32) | (1ull << 63);
    if (key != run_key) {
      flush();
      run_key = key;
      run_cnt = 0;
    }
    ++run_cnt;
  }
  flush();
}
}  // namespace dbcsr_amd
#include "mm_dma.h"
#include "mm_tile_index.h"
namespace dbcsr_amd {

// ---- plan reuse ------------------------------------------------------------------------------------------------------
// A multiply whose operands have the SAME index arrays (patterns, block sizes, block offsets) as the previous multiply of the
// engine -- every SCF step of a CP2K run, every repetition of the performance driver -- needs no new symbolic phase: the engine
// keeps device copies of the last call's index arrays and compares the incoming ones word by word (one small kernel, one flag).
struct PlanSegs {
  const int32_t* a[12];
  const int32_t* b[12];
  long long n[12];  // 32-bit words per segment
  int nseg;
};
__global__ void __launch_bounds__(256) plan_compare(PlanSegs S, int* __restrict__ differs) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  bool bad = false;
  for (int g = 0; g < S.nseg; ++g)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < S.n[g]; i += stride) bad |= S.a[g][i] != S.b[g][i];
  if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(differs, 1);
}

// ----------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------
// Exact-size kernels are instantiated for every cube from 9 to 32 (the reference compiles one kernel per (m, n, k) at run
// time; here the list is fixed at build time and every other case -- mixed sizes, blocks above 32 -- runs the generic kernels; measured on 4 x 4 blocks the generic kernel is 7 % faster, so sizes
// up to 8 are left to it).
#define DBCSR_AMD_HOT_SIZES(X) \
  X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)

#define DBCSR_AMD_DMA_SIZES(X) X(13) X(16) X(23) X(32)

// one wave per C block, only the blocks that are NOT m x n: their squared Frobenius norm (the exact-size kernel wrote the others')
__global__ void __launch_bounds__(256) block_norms_other_sizes(const Desc* __restrict__ descs, int64_t nblk, const double* __restrict__ c_data,
                                                               int m, int n, double* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t cb = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (cb >= nblk) return;
  const Desc d = descs[cb];
  if (d.m == m && d.n == n) return;
  const double* x = c_data + d.c_off;
  double ss = 0.0;
  for (int e = lane; e < d.m * d.n; e += 64) ss += x[e] * x[e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
  if (lane == 0) norms[cb] = ss;
}

// the same for a multiply of mixed sizes: the blocks whose (m, n) class had no run-time compiled kernel (class 9 = other sizes, or hiprtc failed)
struct ClassSet {
  int m[3], n[3], jit_mask;
};
__global__ void __launch_bounds__(256) block_norms_unserved_classes(const Desc* __restrict__ descs, int64_t nblk, const double* __restrict__ c_data,
                                                                    ClassSet cs, double* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t cb = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (cb >= nblk) return;
  const Desc d = descs[cb];
  int rm = 3, rn = 3;
#pragma unroll
  for (int q = 2; q >= 0; --q) {
    if (cs.m[q] > 0 && d.m == cs.m[q]) rm = q;
    if (cs.n[q] > 0 && d.n == cs.n[q]) rn = q;
  }
  if (rm < 3 && rn < 3 && ((cs.jit_mask >> (3 * rm + rn)) & 1)) return;  // its class kernel wrote the norm
  const double* x = c_data + d.c_off;
  double ss = 0.0;
  for (int e = lane; e < d.m * d.n; e += 64) ss += x[e] * x[e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
  if (lane == 0) norms[cb] = ss;
}

static bool launch_hot_f64(int m, int n, int k, dim3 grid, size_t lds_bytes, hipStream_t st, const Desc* descs, int64_t nblk,
                           const Entry* entries, const double* a_data, const double* b_data, double* c_out, const double* c_in,
                           double alpha, double beta, int lds_a, int lds_wave, int dbg, const int* order, const Work* work, int wg_waves,
                           double* norms, int variant) {
  if (m != n || m != k) return false;
  // profiling variants exist for the benchmark's block size only (ablation switches; unpaired fragment reads)
  if (m == 23 && (variant == 1 || variant == 2)) {
    if (variant == 1)
      hipLaunchKernelGGL((mm_numeric_f64_hot<23, 23, 23, 1>), grid, dim3(64 * wg_waves), lds_bytes, st, descs, nblk, entries, a_data, b_data, c_out,
                         c_in, alpha, beta, lds_a, lds_wave, dbg, order, work, norms);
    else
      hipLaunchKernelGGL((mm_numeric_f64_hot<23, 23, 23, 2>), grid, dim3(64 * wg_waves), lds_bytes, st, descs, nblk, entries, a_data, b_data, c_out,
                         c_in, alpha, beta, lds_a, lds_wave, dbg, order, work, norms);
    return true;
  }
  switch (m) {
#define DBCSR_HOT_CASE(S_)                                                                                                      \
  case S_:                                                                                                                      \
    hipLaunchKernelGGL((mm_numeric_f64_hot<S_, S_, S_, 0>), grid, dim3(64 * wg_waves), lds_bytes, st, descs, nblk, entries, a_data, b_data, c_out, \
                       c_in, alpha, beta, lds_a, lds_wave, dbg, order, work, norms);                                            \
    return true;
    DBCSR_AMD_HOT_SIZES(DBCSR_HOT_CASE)
#undef DBCSR_HOT_CASE
    default: return false;
  }
}

// LDS-DMA variant of the exact-size kernel (mm_dma.h): S ring slots per wave, one wave per workgroup
template <int S_>
static bool launch_dma_f64_s(int m, int n, int k, unsigned npos, hipStream_t st, const Desc* descs, int64_t nblk, const Entry* entries,
                             const double* a_data, const double* b_data, double* c_out, const double* c_in, double alpha, double beta,
                             int skip_empty, const int* order) {
  if (m != n || m != k) return false;
  switch (m) {
#define DBCSR_DMA_CASE(S__)                                                                                                   \
  case S__:                                                                                                                   \
    hipLaunchKernelGGL((mm_numeric_f64_dma<S__, S__, S__, S_>), dim3(npos), dim3(64), (DmaRing<S__, S__, S__, S_>::BYTES), st, descs, \
                       nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order);                           \
    return true;
    DBCSR_AMD_DMA_SIZES(DBCSR_DMA_CASE)
#undef DBCSR_DMA_CASE
    default: return false;
  }
}
static bool launch_dma_f64(int S, int m, int n, int k, unsigned npos, hipStream_t st, const Desc* descs, int64_t nblk,
                           const Entry* entries, const double* a_data, const double* b_data, double* c_out, const double* c_in,
                           double alpha, double beta, int skip_empty, const int* order) {
  switch (S) {
    case 2: return launch_dma_f64_s<2>(m, n, k, npos, st, descs, nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order);
    case 3: return launch_dma_f64_s<3>(m, n, k, npos, st, descs, nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order);
    case 4: return launch_dma_f64_s<4>(m, n, k, npos, st, descs, nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order);
    default: return false;
  }
}

static bool launch_hot_f32(int m, int n, int k, dim3 grid, int wg_waves, hipStream_t st, const Desc* descs, int64_t nblk, const Entry* entries,
                           const float* a_data, const float* b_data, float* c_out, const float* c_in, float alpha, float beta,
                           int skip_empty, const int* order) {
  if (m != n || m != k) return false;
  switch (m) {
#define DBCSR_HOT_CASE(S_)                                                                                                     \
  case S_:                                                                                                                     \
    hipLaunchKernelGGL((mm_numeric_f32_hot<S_, S_, S_>), grid, dim3(64 * wg_waves), f32_lds_bytes(wg_waves), st, descs, nblk, entries, a_data, b_data, c_out, c_in, \
                       alpha, beta, skip_empty, order);                                                                        \
    return true;
    DBCSR_AMD_HOT_SIZES(DBCSR_HOT_CASE)
#undef DBCSR_HOT_CASE
    default: return false;
  }
}

struct Engine {
  DevBuf<uint32_t> b_bm, c_bm, cin_bm;
  DevBuf<int> b_pre, c_pre, cin_pre, row_nnz, prod_cnt, blk_nze, tmp_i32;
  DevBuf<int64_t> prod_start, c_blk_p_ws, partial, off_a, off_b;
  DevBuf<Entry> entries;
  DevBuf<Desc> descs;
  DevBuf<double> row_sums, norms64;
  DevBuf<float> a_norms, b_norms;
  DevBuf<int> keep;
  FilterArgs filter = {nullptr, nullptr, 0.0f};
  int64_t flt_nblks = 0;
  DevBuf<int> order, order_cnt;
  const void* norms_data = nullptr;  // norms64[] holds the block norms of the matrix with this data pointer (left by the numeric kernel)
  int64_t norms_nblks = 0;
  int canonical_c = 0;  // dbcsr_amd_mm_set_canonical_product: the product matrix has symmetry, its index is in canonical form
  int wg_waves = 0;   // DBCSR_AMD_MM_WG_WAVES = 1 | 2 | 4: waves per workgroup of the one-wave-per-C-block kernels (0: by list length).  A workgroup's LDS is
                      // released when its LAST wave ends, so with product lists of uneven length fewer waves per workgroup keep
                      // more of the CU's wave slots busy (config 3: kernel 8.93 / 8.09 / 7.51 ms for 4 / 2 / 1, config 2: 23.6 / 22.7 /
                      // 22.6, config 4: 30.1 / 28.8 / 28.4 on the same box, profiles/r02_wg_waves_bench_lines.txt)
  DevBuf<Work> work;  // launch-order records of the exact-size fp64 kernels (DBCSR_AMD_MM_WORK=0: the class kernels read order[] -> descs[] -> entries[] instead)
  int use_work = 1;
  DevBuf<int64_t> order_base;
  int64_t order_len = 0;
  Window crop_win = {0, 0, 0, 0};       // window of the last dbcsr_amd_bcsr_crop_count
  bool crop_pending = false;
  int hot_m = 0, hot_n = 0, hot_k = 0;  // dominant block sizes of the last symbolic phase (0: none)
  int use_tiny = 1;                     // DBCSR_AMD_MM_TINY=0: no packed kernel for blocks of at most 4 x 4
  int use_hot = 1;                      // DBCSR_AMD_MM_HOT=0: never use the exact-size kernels
  int lds_pad = 0;                      // DBCSR_AMD_MM_LDS_PAD: extra LDS bytes per workgroup (occupancy experiments)
  int64_t panel_bytes = 160ll << 20;  // DBCSR_AMD_MM_PANEL_MB: target size of a B column panel
  int row_group = 0;                 // DBCSR_AMD_MM_ROW_GROUP: rows walked together per XCD (0 = automatic)
  DevBuf<unsigned long long> dev_scalars, stat_table;
  int64_t* host_scalars = nullptr;  // pinned: [0]=c_nblks [1]=c_nze [2]=nproducts [3]=flop
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // around fill_products and the numeric kernel
  bool timed = false;
  // state carried from symbolic to numeric
  int nbr = 0, W = 0;
  int64_t c_nblks = 0, nproducts = 0;
  bool have_cin = false, retain = false, valid = false;
  int max_m = 0, max_k = 0, max_n = 0, min_m = 0, min_k = 0, min_n = 0;
  bool rows_kernels = false;  // product-driven symbolic kernels (sparse C); DBCSR_AMD_MM_SYMBOLIC=rows forces, =grid / =word exclude
  int force_symbolic = 0;     // 0 automatic, 1 word, 2 grid, 3 rows
  bool grid_kernels = false, force_word_kernels = false;  // DBCSR_AMD_MM_SYMBOLIC=word forces the per-word symbolic kernels
  int dbg = 0;      // DBCSR_AMD_MM_DBG: ablation switches of the LDS kernel (profiling only; the exact-size kernel honours them in its VAR = 1 build)
  // XCD-wide C tiles in registers (mm_tile.h): DBCSR_AMD_MM_TILE = 0 never, 1 automatic, 2 whenever the sizes allow;
  // DBCSR_AMD_MM_TILE_WINDOW = k window of the team (inner blocks; 0: no throttle); DBCSR_AMD_MM_TILE_RDV = 1: unpaired fragment reads
  int use_tile = 0, tile_window = 256, tile_rdv = 0, tile_pub = 1, tile_prefetch = 0;  // DBCSR_AMD_MM_TILE_PUB: progress stores written through (0) / left in L2 (1)
  int hot_cnt_m = 0, hot_cnt_k = 0, hot_cnt_n = 0;  // block rows / inner blocks / block columns of the dominant size
  DevBuf<uint32_t> a_bm, bt_bm, tile_prog;
  DevBuf<int> a_pre, tile_rows, tile_cols, tile_cnt, tile_flags;
  DevBuf<int64_t> tile_start;
  DevBuf<TileDesc> tdescs;
  DevBuf<TileEntry> tentries;
  // plan reuse (plan_compare): device copies of the index arrays the last symbolic phase saw, C's index as the numeric phase emitted it
  int use_plan = 1;  // DBCSR_AMD_MM_PLAN=0: every multiply runs its symbolic phase
  bool plan_saved = false, plan_hit = false, plan_numeric = false;
  int plan_dims[3] = {0, 0, 0}, plan_retain = 0, plan_canonical = 0, plan_datatype = 0;
  int64_t plan_nblks[3] = {0, 0, 0};
  const void* plan_ptrs[12] = {nullptr};
  DevBuf<int32_t> plan_words, plan_c_col_i;
  DevBuf<int64_t> plan_c_blk_p;
  DevBuf<int> plan_flag;
  int* plan_host_flag = nullptr;  // pinned
  dbcsr_amd_mm_counts plan_counts = {0, 0, 0, 0};
  bool work_built = false, tile_built = false;
  TileGeom tile_geom = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long plan_hits = 0, plan_misses = 0;
  int hot_variant = 0;  // DBCSR_AMD_MM_HOT_VARIANT: 2 = exact-size kernel with unpaired ds_read_b64 fragment reads (23^3 only)
  int use_pipe = -1, pipe_g = 8;  // multi-block pipelined kernel: -1 automatic (short product lists only), DBCSR_AMD_MM_KERNEL=pipe|lds1 forces; DBCSR_AMD_MM_PIPE_G = blocks per wave
  // (m, n) classes (mixed block sizes, see order_count_cls): DBCSR_AMD_MM_CLASSES = 0 never, 1 automatic, 2 always when the sizes allow
  int use_classes = 1;
  int class_g = 1;  // DBCSR_AMD_MM_CLASS_G: C blocks per wave in the class kernels (1, 2, 4, 8)
  bool cls_mode = false;
  int cls_m[3] = {0, 0, 0}, cls_n[3] = {0, 0, 0}, cls_k[3] = {0, 0, 0};
  int64_t cls_len[kNumClasses] = {0}, cls_off[kNumClasses] = {0};
  DevBuf<int> cls_hist;
  DevBuf<unsigned char> cls_row, cls_col;
  DevBuf<uint32_t> cls_col_bm;
  DevBuf<int64_t> cls_lens;
  int* cls_host_hist = nullptr;       // pinned: 3 x 33 size histograms
  int64_t* cls_host_lens = nullptr;   // pinned: 10 lengths, 10 offsets, total
  char last_kernel[96] = "";  // name of the numeric kernel of the last dbcsr_amd_mm_numeric (dbcsr_amd_mm_last_kernel)
  int dma_stages = 0;  // DBCSR_AMD_MM_KERNEL=dma2|dma3|dma4: LDS-DMA exact-size kernel with that many ring slots (0: off)
  int use_lds = 1;  // DBCSR_AMD_MM_KERNEL=direct selects the v1 kernel (A/B experiments)
};

// waves per block row for the kernels that stream whole blocks (norms, compaction): enough waves to keep the memory system busy
static inline int row_split(int64_t nbr, int64_t nblks) {
  if (nbr <= 0) return 1;
  const int64_t per_row = nblks / nbr;
  int64_t S = (65536 + nbr - 1) / nbr;
  if (S > per_row) S = per_row;
  return (int)std::max<int64_t>(1, std::min<int64_t>(S, 64));
}

template <typename TO>
static int exclusive_scan(Engine* E, const int* in, int64_t n, TO* out, int64_t* total_dev, bool write_total_at_n, hipStream_t st) {
  const int nb = (int)((n + kScanChunk - 1) / kScanChunk);
  if (E->partial.ensure((size_t)(nb > 0 ? nb : 1))) return -1;
  if (n <= 0) {
    if (total_dev) ACC_CHECK(hipMemsetAsync(total_dev, 0, sizeof(int64_t), st));
    if (write_total_at_n) ACC_CHECK(hipMemsetAsync(out, 0, sizeof(TO), st));
    return 0;
  }
  hipLaunchKernelGGL(scan_reduce, dim3(nb), dim3(kScanThreads), 0, st, in, n, E->partial.p);
  hipLaunchKernelGGL(scan_partials, dim3(1), dim3(kScanThreads), 0, st, E->partial.p, nb, total_dev);
  hipLaunchKernelGGL((scan_apply<TO>), dim3(nb), dim3(kScanThreads), 0, st, in, n, E->partial.p, out, write_total_at_n ? 1 : 0);
  return check(hipGetLastError(), "exclusive_scan", __FILE__, __LINE__);
}

static inline dim3 grid_for(int64_t nthreads) { return dim3((unsigned)((nthreads + 255) / 256)); }

static inline void plan_invalidate(Engine* E) { E->plan_saved = E->plan_hit = E->plan_numeric = false; }

// the twelve index arrays a plan depends on, as 32-bit words: patterns, block offsets and block sizes of A, B, C_in
static void plan_segments(const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in, const void* (&ptr)[12], long long (&n)[12]) {
  const void* p[12] = {a->row_p, a->col_i, a->blk_p, b->row_p, b->col_i, b->blk_p, c_in->row_p, c_in->col_i, c_in->blk_p,
                       a->row_blk_size, a->col_blk_size, b->col_blk_size};
  const long long w[12] = {a->nblkrows + 1ll, a->nblks, 2 * a->nblks, b->nblkrows + 1ll, b->nblks, 2 * b->nblks, c_in->nblkrows + 1ll, c_in->nblks,
                           2 * c_in->nblks, a->nblkrows, a->nblkcols, b->nblkcols};
  for (int i = 0; i < 12; ++i) ptr[i] = p[i], n[i] = w[i];
}

// 1 = the operands have exactly the index arrays of the saved plan (synchronises the stream once), 0 = not, < 0 error
static int plan_matches(Engine* E, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in, int retain, hipStream_t st) {
  if (!E->use_plan || !E->plan_saved) return 0;
  if (a->nblkrows != E->plan_dims[0] || a->nblkcols != E->plan_dims[1] || b->nblkcols != E->plan_dims[2] || a->nblks != E->plan_nblks[0] ||
      b->nblks != E->plan_nblks[1] || c_in->nblks != E->plan_nblks[2] || retain != E->plan_retain || E->canonical_c != E->plan_canonical)
    return 0;
  const void* ptr[12];
  long long n[12];
  plan_segments(a, b, c_in, ptr, n);
  PlanSegs S;
  S.nseg = 12;
  long long off = 0, total = 0;
  for (int i = 0; i < 12; ++i) {
    S.a[i] = static_cast<const int32_t*>(ptr[i]);
    S.b[i] = E->plan_words.p + off;
    S.n[i] = n[i];
    off += n[i];
    total += n[i];
  }
  ACC_CHECK(hipMemsetAsync(E->plan_flag.p, 0, sizeof(int), st));
  const unsigned nb = (unsigned)std::min<long long>(2048, std::max<long long>(1, (total / 12 + 255) / 256));
  hipLaunchKernelGGL(plan_compare, dim3(nb), dim3(256), 0, st, S, E->plan_flag.p);
  ACC_CHECK(hipMemcpyAsync(E->plan_host_flag, E->plan_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  return *E->plan_host_flag == 0 ? 1 : 0;
}

// keep device copies of the index arrays this symbolic phase saw, and of C's row pointer
static int plan_save(Engine* E, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in, int retain, const int32_t* c_row_p,
                     const dbcsr_amd_mm_counts& counts, hipStream_t st) {
  plan_invalidate(E);
  if (!E->use_plan) return 0;
  const void* ptr[12];
  long long n[12];
  plan_segments(a, b, c_in, ptr, n);
  long long total = 0;
  for (int i = 0; i < 12; ++i) total += n[i];
  const long long crow = a->nblkrows + 1ll;
  if (E->plan_words.ensure((size_t)(total + crow) + 1) || E->plan_flag.ensure(4)) return -1;
  long long off = 0;
  for (int i = 0; i < 12; ++i) {
    if (n[i] > 0) ACC_CHECK(hipMemcpyAsync(E->plan_words.p + off, ptr[i], sizeof(int32_t) * (size_t)n[i], hipMemcpyDeviceToDevice, st));
    off += n[i];
  }
  ACC_CHECK(hipMemcpyAsync(E->plan_words.p + off, c_row_p, sizeof(int32_t) * (size_t)crow, hipMemcpyDeviceToDevice, st));
  E->plan_dims[0] = a->nblkrows, E->plan_dims[1] = a->nblkcols, E->plan_dims[2] = b->nblkcols;
  E->plan_nblks[0] = a->nblks, E->plan_nblks[1] = b->nblks, E->plan_nblks[2] = c_in->nblks;
  E->plan_retain = retain;
  E->plan_canonical = E->canonical_c;
  E->plan_counts = counts;
  E->plan_saved = true;
  return 0;
}

// The tile dataflow (mm_tile.h) for the C blocks of the dominant size: index work (bitmaps of A and of B transposed, sub-tile
// descriptors, k-sorted product lists), the persistent tile kernel, the products with inner blocks of another size.  The caller
// then runs the exact-size kernel over the C blocks of the other sizes.  descs[] and C_out's index are already filled.
template <int S_>
static int run_tile_f64(Engine* E, hipStream_t st, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
                        dbcsr_amd_bcsr* c_out, double alpha, double beta) {
  const int nbr = a->nblkrows, nbk = a->nblkcols, nbc = b->nblkcols, W = E->W, Wk = (nbk + 31) / 32;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    ACC_CHECK(hipGetDevice(&dev));
    ACC_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int cu_per_xcd = std::min(32, std::max(2, n_cu / 8));
  TileGeom G;
  G.nfr = E->hot_cnt_m;
  G.nfc = E->hot_cnt_n;
  G.nTR = (G.nfr + kTileT - 1) / kTileT;
  G.nTC = (G.nfc + kTileT - 1) / kTileT;
  G.team_rows = std::max(1, cu_per_xcd * 8 / kTeamCols);
  G.nSR = (G.nTR + G.team_rows - 1) / G.team_rows;
  G.nSC = (G.nTC + kTeamCols - 1) / kTeamCols;
  G.nseq = (G.nSR * G.nSC + 7) / 8;
  G.kspan = nbk + 1;
  if ((int64_t)G.nseq * G.kspan >= 0x7ff00000ll) return 1;  // progress counter would overflow: not a tile case
  const int64_t nT = (int64_t)G.nTR * G.nTC;
  const bool reuse = E->plan_hit && E->plan_numeric && E->tile_built;
  if (E->tile_prog.ensure(8 * 256) || E->tile_flags.ensure(4)) return -1;
  ACC_CHECK(hipMemsetAsync(E->tile_prog.p, 0, sizeof(uint32_t) * 8 * 256, st));
  ACC_CHECK(hipMemsetAsync(E->tile_flags.p, 0, sizeof(int) * 4, st));
  if (!reuse) {
  if (E->a_bm.ensure((size_t)nbr * Wk + 1) || E->a_pre.ensure((size_t)nbr * Wk + 1) || E->bt_bm.ensure((size_t)nbc * Wk + 1) ||
      E->tile_rows.ensure((size_t)nbr + 1) || E->tile_cols.ensure((size_t)nbc + 1) || E->tdescs.ensure((size_t)nT + 1) ||
      E->tile_cnt.ensure((size_t)nT + 1) || E->tile_start.ensure((size_t)nT + 1) || E->tentries.ensure((size_t)E->nproducts + 1) ||
      false)
    return -1;
  ACC_CHECK(hipMemsetAsync(E->a_bm.p, 0, sizeof(uint32_t) * (size_t)nbr * Wk, st));
  ACC_CHECK(hipMemsetAsync(E->bt_bm.p, 0, sizeof(uint32_t) * (size_t)nbc * Wk, st));
  hipLaunchKernelGGL(bitmap_from_index, grid_for((int64_t)nbr * 64), dim3(256), 0, st, a->row_p, a->col_i, nbr, Wk, E->a_bm.p);
  hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->a_bm.p, nbr, Wk, E->a_pre.p, (int*)nullptr);
  hipLaunchKernelGGL(tile_bitmap_transposed, grid_for((int64_t)nbk * 64), dim3(256), 0, st, b->row_p, b->col_i, nbk, Wk, E->bt_bm.p);
  hipLaunchKernelGGL(tile_select, dim3(1), dim3(64), 0, st, a->row_blk_size, nbr, S_, E->tile_rows.p, nbr);
  hipLaunchKernelGGL(tile_select, dim3(1), dim3(64), 0, st, b->col_blk_size, nbc, S_, E->tile_cols.p, nbc);
  hipLaunchKernelGGL(tile_descs, grid_for(nT * 16), dim3(256), 0, st, G, E->tile_rows.p, E->tile_cols.p, E->c_bm.p, E->c_pre.p, c_out->row_p, W,
                     E->descs.p, E->tdescs.p, E->tile_cnt.p);
  if (exclusive_scan<int64_t>(E, E->tile_cnt.p, nT, E->tile_start.p, nullptr, false, st)) return -1;
  hipLaunchKernelGGL(tile_lists, grid_for(nT * 64), dim3(256), 0, st, G, E->tile_rows.p, E->tile_cols.p, nbk, Wk, E->a_bm.p, E->a_pre.p, a->row_p,
                     a->blk_p, E->bt_bm.p, W, E->b_bm.p, E->b_pre.p, b->row_p, b->blk_p, a->col_blk_size, S_, E->tile_start.p, E->tile_cnt.p,
                     E->tdescs.p, E->tentries.p, E->tile_flags.p + 1);
  E->tile_built = true;
  }
  TileArgs P;
  P.tdescs = E->tdescs.p;
  P.entries = E->tentries.p;
  P.a_data = static_cast<const double*>(a->data);
  P.b_data = static_cast<const double*>(b->data);
  P.c_out = static_cast<double*>(c_out->data);
  P.c_in = static_cast<const double*>(c_in->data);
  P.alpha = alpha;
  P.beta = beta;
  P.prog = E->tile_prog.p;
  P.flags = E->tile_flags.p;
  P.G = G;
  P.window = E->tile_window;
  P.pub_policy = E->tile_pub;
  P.prefetch = E->tile_prefetch;
  ACC_CHECK(hipEventRecord(E->ev[1], st));  // the timed numeric launch starts here (the index work above counts as fill time)
  if (tile_launch(S_, S_, S_, E->tile_rdv, (unsigned)(8 * cu_per_xcd), st, P)) return -1;
  if (tile_launch_remainder(S_, S_, st, G, E->tdescs.p, E->tentries.p, P.a_data, P.b_data, P.c_out, alpha)) return -1;
  return check(hipGetLastError(), "run_tile_f64", __FILE__, __LINE__);
}

}  // namespace dbcsr_amd

using namespace dbcsr_amd;

extern "C" {

int dbcsr_amd_mm_create(void** handle) {
  if (!handle) return -1;
  Engine* E = new (std::nothrow) Engine();
  if (!E) return -1;
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&E->host_scalars), 16 * sizeof(int64_t), hipHostMallocDefault);
  if (e != hipSuccess) {
    delete E;
    return check(e, "hipHostMalloc", __FILE__, __LINE__);
  }
  if (const char* k = getenv("DBCSR_AMD_MM_KERNEL")) {
    E->use_lds = strcmp(k, "direct") != 0;
    E->use_pipe = strcmp(k, "pipe") == 0 ? 1 : (strcmp(k, "lds1") == 0 ? 0 : -1);
    if (strncmp(k, "dma", 3) == 0 && k[3] >= '2' && k[3] <= '4') E->dma_stages = k[3] - '0';
  }
  if (const char* k = getenv("DBCSR_AMD_MM_PIPE_G")) E->pipe_g = std::max(1, atoi(k));
  if (const char* k = getenv("DBCSR_AMD_MM_CLASSES")) E->use_classes = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_WORK")) E->use_work = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_WG_WAVES")) {
    const int w = atoi(k);
    if (w == 1 || w == 2 || w == 4) E->wg_waves = w;
  }
  if (const char* k = getenv("DBCSR_AMD_MM_CLASS_G")) {
    const int g = atoi(k);
    E->class_g = (g == 2 || g == 4 || g == 8) ? g : 1;
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&E->cls_host_hist), 3 * 33 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&E->cls_host_lens), (2 * kNumClasses + 1) * sizeof(int64_t), hipHostMallocDefault) != hipSuccess)
    return -1;
  if (const char* k = getenv("DBCSR_AMD_MM_DBG")) E->dbg = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_HOT_VARIANT")) E->hot_variant = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_PLAN")) E->use_plan = atoi(k);
  if (hipHostMalloc(reinterpret_cast<void**>(&E->plan_host_flag), sizeof(int), hipHostMallocDefault) != hipSuccess) return -1;
  if (const char* k = getenv("DBCSR_AMD_MM_TILE")) E->use_tile = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_WINDOW")) E->tile_window = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_RDV")) E->tile_rdv = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_PUB")) E->tile_pub = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_PREFETCH")) E->tile_prefetch = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_HOT")) E->use_hot = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TINY")) E->use_tiny = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_LDS_PAD")) E->lds_pad = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_SYMBOLIC")) {
    E->force_word_kernels = strcmp(k, "word") == 0;
    E->force_symbolic = strcmp(k, "word") == 0 ? 1 : (strcmp(k, "grid") == 0 ? 2 : (strcmp(k, "rows") == 0 ? 3 : 0));
  }
  if (const char* k = getenv("DBCSR_AMD_MM_PANEL_MB")) E->panel_bytes = (int64_t)atoll(k) << 20;
  if (const char* k = getenv("DBCSR_AMD_MM_ROW_GROUP")) E->row_group = atoi(k);
  for (int i = 0; i < 3; ++i) {
    e = hipEventCreate(&E->ev[i]);
    if (e != hipSuccess) return check(e, "hipEventCreate", __FILE__, __LINE__);
  }
  *handle = E;
  return 0;
}

int dbcsr_amd_mm_timing(void* handle, float* ms_fill, float* ms_numeric) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !E->timed) return -1;
  ACC_CHECK(hipEventSynchronize(E->ev[2]));
  float f = 0.f, n = 0.f;
  ACC_CHECK(hipEventElapsedTime(&f, E->ev[0], E->ev[1]));
  ACC_CHECK(hipEventElapsedTime(&n, E->ev[1], E->ev[2]));
  if (ms_fill) *ms_fill = f;
  if (ms_numeric) *ms_numeric = n;
  return 0;
}

int dbcsr_amd_mm_destroy(void* handle) {
  if (!handle) return 0;
  Engine* E = static_cast<Engine*>(handle);
  E->b_bm.release(); E->c_bm.release(); E->cin_bm.release();
  E->b_pre.release(); E->c_pre.release(); E->cin_pre.release(); E->row_nnz.release(); E->prod_cnt.release();
  E->blk_nze.release(); E->tmp_i32.release();
  E->prod_start.release(); E->c_blk_p_ws.release(); E->partial.release(); E->off_a.release(); E->off_b.release();
  E->entries.release(); E->descs.release(); E->row_sums.release(); E->dev_scalars.release();
  E->order.release(); E->order_cnt.release(); E->order_base.release();
  E->stat_table.release();
  E->norms64.release(); E->a_norms.release(); E->b_norms.release(); E->keep.release();
  E->a_bm.release(); E->bt_bm.release(); E->tile_prog.release(); E->a_pre.release(); E->tile_rows.release(); E->tile_cols.release();
  E->tile_cnt.release(); E->tile_flags.release(); E->tile_start.release(); E->tdescs.release(); E->tentries.release();
  if (E->host_scalars) (void)hipHostFree(E->host_scalars);
  if (E->plan_host_flag) (void)hipHostFree(E->plan_host_flag);
  E->plan_words.release(); E->plan_c_col_i.release(); E->plan_c_blk_p.release(); E->plan_flag.release(); E->work.release();
  if (E->cls_host_hist) (void)hipHostFree(E->cls_host_hist);
  if (E->cls_host_lens) (void)hipHostFree(E->cls_host_lens);
  E->cls_hist.release(); E->cls_row.release(); E->cls_col.release(); E->cls_col_bm.release(); E->cls_lens.release();
  for (int i = 0; i < 3; ++i)
    if (E->ev[i]) (void)hipEventDestroy(E->ev[i]);
  delete E;
  return 0;
}

int dbcsr_amd_mm_symbolic(void* handle, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
                          int retain_sparsity, int32_t* c_out_row_p, dbcsr_amd_mm_counts* counts, void* stream) {
  return dbcsr_amd_mm_symbolic_filtered(handle, dbcsr_type_real_8, 1.0, 0.0, a, b, c_in, retain_sparsity, c_out_row_p, counts, stream);
}

int dbcsr_amd_mm_symbolic_filtered(void* handle, libsmm_acc_data_t datatype, double alpha, double filter_eps, const dbcsr_amd_bcsr* a,
                                   const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in, int retain_sparsity, int32_t* c_out_row_p,
                                   dbcsr_amd_mm_counts* counts, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !a || !b || !c_in || !c_out_row_p || !counts) return -1;
  const bool filtering = filter_eps > 0.0;
  if (filtering && datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  if (a->nblkcols != b->nblkrows || a->nblkrows != c_in->nblkrows || b->nblkcols != c_in->nblkcols) {
    fprintf(stderr, "dbcsr_amd_mm_symbolic: incompatible block dimensions\n");
    return -2;
  }
  hipStream_t st = stream_of(stream);
  const int nbr = a->nblkrows, nbk = a->nblkcols, nbc = b->nblkcols;
  const int W = (nbc + 31) / 32;
  // same index arrays as the previous multiply of this engine: its plan stands (no on-the-fly filter: that one depends on the values)
  if (!filtering) {
    const int hit = plan_matches(E, a, b, c_in, retain_sparsity ? 1 : 0, st);
    if (hit < 0) return -1;
    if (hit) {
      long long off = 0;
      const void* ptr[12];
      long long n[12];
      plan_segments(a, b, c_in, ptr, n);
      for (int i = 0; i < 12; ++i) off += n[i];
      ACC_CHECK(hipMemcpyAsync(c_out_row_p, E->plan_words.p + off, sizeof(int32_t) * ((size_t)nbr + 1), hipMemcpyDeviceToDevice, st));
      *counts = E->plan_counts;
      E->norms_data = nullptr;
      E->filter = FilterArgs{nullptr, nullptr, 0.0f};
      E->valid = true;
      E->plan_hit = true;
      ++E->plan_hits;
      return 0;
    }
  }
  plan_invalidate(E);
  E->valid = false;
  E->nbr = nbr;
  E->W = W;
  E->retain = retain_sparsity != 0;
  E->norms_data = nullptr;  // block norms left by an earlier numeric phase belong to that product only
  E->have_cin = c_in->nblks > 0;
  if (E->b_bm.ensure((size_t)nbk * W + 1) || E->b_pre.ensure((size_t)nbk * W + 1) || E->c_bm.ensure((size_t)nbr * W + 1) ||
      E->c_pre.ensure((size_t)nbr * W + 1) || E->row_nnz.ensure((size_t)nbr + 1) || E->dev_scalars.ensure(16))
    return -1;
  if (E->have_cin && (E->cin_bm.ensure((size_t)nbr * W + 1) || E->cin_pre.ensure((size_t)nbr * W + 1))) return -1;
  ACC_CHECK(hipMemsetAsync(E->dev_scalars.p, 0, 16 * sizeof(unsigned long long), st));
  {  // the "negated min" slots start at the most negative value
    static const int init[6] = {0, -0x7fffffff, 0, -0x7fffffff, 0, -0x7fffffff};
    ACC_CHECK(hipMemcpyAsync(E->dev_scalars.p + 4, init, sizeof(init), hipMemcpyHostToDevice, st));
  }
  if (nbr == 0 || nbc == 0) {
    ACC_CHECK(hipMemsetAsync(c_out_row_p, 0, sizeof(int32_t) * ((size_t)nbr + 1), st));
    ACC_CHECK(hipStreamSynchronize(st));
    counts->c_nblks = counts->c_nze = counts->nproducts = counts->flop = 0;
    E->c_nblks = 0;
    E->nproducts = 0;
    E->valid = true;
    return 0;
  }
  // 1. bitmaps of B (and C_in)
  ACC_CHECK(hipMemsetAsync(E->b_bm.p, 0, sizeof(uint32_t) * (size_t)nbk * W, st));
  if (nbk > 0) {
    hipLaunchKernelGGL(bitmap_from_index, grid_for((int64_t)nbk * 64), dim3(256), 0, st, b->row_p, b->col_i, nbk, W, E->b_bm.p);
    hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbk * 64), dim3(256), 0, st, E->b_bm.p, nbk, W, E->b_pre.p, (int*)nullptr);
  }
  if (E->have_cin) {
    ACC_CHECK(hipMemsetAsync(E->cin_bm.p, 0, sizeof(uint32_t) * (size_t)nbr * W, st));
    hipLaunchKernelGGL(bitmap_from_index, grid_for((int64_t)nbr * 64), dim3(256), 0, st, c_in->row_p, c_in->col_i, nbr, W,
                       E->cin_bm.p);
    hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->cin_bm.p, nbr, W, E->cin_pre.p, (int*)nullptr);
  }
  // on-the-fly filter: block norms of A and alpha*B (fp32 values of fp64 sums)
  E->filter = FilterArgs{nullptr, nullptr, 0.0f};
  if (filtering) {
    if (E->a_norms.ensure((size_t)a->nblks + 1) || E->b_norms.ensure((size_t)b->nblks + 1)) return -1;
    const int sa = row_split(nbr, a->nblks), sb = row_split(nbk, b->nblks);
    if (datatype == dbcsr_type_real_8) {
      hipLaunchKernelGGL((bcsr_block_norms<double>), grid_for((int64_t)nbr * sa * 64), dim3(256), 0, st, a->row_p, a->col_i, a->blk_p,
                         static_cast<const double*>(a->data), a->row_blk_size, a->col_blk_size, nbr, sa, 1.0, E->a_norms.p, (double*)nullptr);
      hipLaunchKernelGGL((bcsr_block_norms<double>), grid_for((int64_t)nbk * sb * 64), dim3(256), 0, st, b->row_p, b->col_i, b->blk_p,
                         static_cast<const double*>(b->data), b->row_blk_size, b->col_blk_size, nbk, sb, alpha, E->b_norms.p, (double*)nullptr);
    } else {
      hipLaunchKernelGGL((bcsr_block_norms<float>), grid_for((int64_t)nbr * sa * 64), dim3(256), 0, st, a->row_p, a->col_i, a->blk_p,
                         static_cast<const float*>(a->data), a->row_blk_size, a->col_blk_size, nbr, sa, 1.0, E->a_norms.p, (double*)nullptr);
      hipLaunchKernelGGL((bcsr_block_norms<float>), grid_for((int64_t)nbk * sb * 64), dim3(256), 0, st, b->row_p, b->col_i, b->blk_p,
                         static_cast<const float*>(b->data), b->row_blk_size, b->col_blk_size, nbk, sb, alpha, E->b_norms.p, (double*)nullptr);
    }
    E->filter = FilterArgs{E->a_norms.p, E->b_norms.p, (float)filter_eps};
  }
  // 2. pattern of C_out, its row prefix and row pointer
  // expected number of products against the number of (row, column) candidates: product-driven kernels for a sparse product
  const double prod_est = (double)a->nblks * ((double)b->nblks / (double)std::max(nbk, 1));
  const bool sparse_guess = E->force_symbolic == 3 || (E->force_symbolic == 0 && nbr >= 2048 && prod_est < 0.6 * (double)nbr * (double)nbc);
  if (filtering && sparse_guess && !retain_sparsity) {
    if (E->have_cin)
      ACC_CHECK(hipMemcpyAsync(E->c_bm.p, E->cin_bm.p, sizeof(uint32_t) * (size_t)nbr * W, hipMemcpyDeviceToDevice, st));
    else
      ACC_CHECK(hipMemsetAsync(E->c_bm.p, 0, sizeof(uint32_t) * (size_t)nbr * W, st));
    hipLaunchKernelGGL(c_bitmap_rows_filtered, grid_for((int64_t)nbr * 64), dim3(256), 0, st, a->row_p, a->col_i, b->row_p, b->col_i, nbr, W,
                       E->canonical_c, E->filter, E->c_bm.p);
  } else if (filtering)
    hipLaunchKernelGGL(c_bitmap_filtered, grid_for((int64_t)nbr * ((nbc + 63) / 64) * 64), dim3(256), 0, st, a->row_p, a->col_i, b->row_p,
                       E->b_bm.p, E->b_pre.p, E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr, nbr, nbc, W, (nbc + 63) / 64,
                       retain_sparsity ? 1 : 0, E->canonical_c, E->filter, E->c_bm.p);
  else
    hipLaunchKernelGGL(c_bitmap, grid_for((int64_t)nbr * W), dim3(256), 0, st, a->row_p, a->col_i, E->b_bm.p,
                       E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr, nbr, W, retain_sparsity ? 1 : 0, E->canonical_c, E->c_bm.p);
  hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->c_bm.p, nbr, W, E->c_pre.p, E->row_nnz.p);
  int64_t* dsc = reinterpret_cast<int64_t*>(E->dev_scalars.p);
  if (exclusive_scan<int32_t>(E, E->row_nnz.p, nbr, c_out_row_p, dsc + 0, true, st)) return -1;
  // block-size maxima (LDS slice size / kernel choice of the numeric phase)
  {
    int* mx = reinterpret_cast<int*>(E->dev_scalars.p + 4);
    hipLaunchKernelGGL(max_of, dim3(1), dim3(256), 0, st, a->row_blk_size, nbr, mx + 0);
    hipLaunchKernelGGL(max_of, dim3(1), dim3(256), 0, st, a->col_blk_size, nbk, mx + 2);
    hipLaunchKernelGGL(max_of, dim3(1), dim3(256), 0, st, b->col_blk_size, nbc, mx + 4);
    // ... and the most frequent block size per dimension (choice of an exact-size kernel)
    int* md = reinterpret_cast<int*>(E->dev_scalars.p + 8);
    hipLaunchKernelGGL(mode_of, dim3(1), dim3(256), 0, st, a->row_blk_size, nbr, md + 0);
    hipLaunchKernelGGL(mode_of, dim3(1), dim3(256), 0, st, a->col_blk_size, nbk, md + 2);
    hipLaunchKernelGGL(mode_of, dim3(1), dim3(256), 0, st, b->col_blk_size, nbc, md + 4);
  }
  // block-size histograms (sizes 1..32) of the three dimensions: the (m, n) classes of a mixed-size multiply
  E->cls_mode = false;
  if (E->use_classes > 0) {
    if (E->cls_hist.ensure(3 * 33)) return -1;
    ACC_CHECK(hipMemsetAsync(E->cls_hist.p, 0, 3 * 33 * sizeof(int), st));
    hipLaunchKernelGGL(size_hist, grid_for(nbr), dim3(256), 0, st, a->row_blk_size, nbr, E->cls_hist.p);
    hipLaunchKernelGGL(size_hist, grid_for(nbc), dim3(256), 0, st, b->col_blk_size, nbc, E->cls_hist.p + 33);
    hipLaunchKernelGGL(size_hist, grid_for(nbk), dim3(256), 0, st, a->col_blk_size, nbk, E->cls_hist.p + 66);
    ACC_CHECK(hipMemcpyAsync(E->cls_host_hist, E->cls_hist.p, 3 * 33 * sizeof(int), hipMemcpyDeviceToHost, st));
  }
  // need c_nblks (and the block-size extrema) on the host to size per-block work arrays
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 11 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  const int64_t c_nblks = E->host_scalars[0];
  {
    const int* mx = reinterpret_cast<const int*>(E->host_scalars + 4);
    E->max_m = mx[0]; E->min_m = -mx[1];
    E->max_k = mx[2]; E->min_k = -mx[3];
    E->max_n = mx[4]; E->min_n = -mx[5];
    if (E->max_k > 0xffff || E->max_m > 0x7fff || E->max_n > 0x7fff) {
      fprintf(stderr, "dbcsr_amd_mm_symbolic: block sizes above 32767 (m, n) / 65535 (k) are not supported (packed 16-bit extents)\n");
      return -1;
    }
    // exact-size kernel: only when one (m, n, k) covers at least 90 % of the block rows / columns of each dimension
    const int* md = reinterpret_cast<const int*>(E->host_scalars + 8);
    const bool dominant = 10ll * md[1] >= 9ll * nbr && 10ll * md[3] >= 9ll * nbk && 10ll * md[5] >= 9ll * nbc;
    E->hot_m = dominant ? md[0] : 0;
    E->hot_k = dominant ? md[2] : 0;
    E->hot_n = dominant ? md[4] : 0;
    E->hot_cnt_m = md[1], E->hot_cnt_k = md[3], E->hot_cnt_n = md[5];
  }
  // (m, n) classes: blocks of at most 32 in every dimension, no single dominant size (that case has its ahead-of-time
  // kernel), not the packed 4 x 4 case, and enough C blocks to pay for compiling the class kernels (forced with
  // DBCSR_AMD_MM_CLASSES=2)
  if (E->use_classes > 0 && E->max_m <= 32 && E->max_k <= 32 && E->max_n <= 32 && E->min_m >= 1 && E->min_k >= 1 && E->min_n >= 1 &&
      !(E->max_m <= 4 && E->max_n <= 4) && (E->use_classes > 1 || (E->hot_m == 0 && c_nblks >= 200000))) {
    auto top3 = [](const int* hist, int* out) {
      int used[3] = {-1, -1, -1};
      for (int r = 0; r < 3; ++r) {
        int best = 0, bc = 0;
        for (int sz = 1; sz <= 32; ++sz)
          if (hist[sz] > bc && sz != used[0] && sz != used[1]) best = sz, bc = hist[sz];
        out[r] = best;
        used[r] = best ? best : -1;
      }
    };
    top3(E->cls_host_hist, E->cls_m);
    top3(E->cls_host_hist + 33, E->cls_n);
    top3(E->cls_host_hist + 66, E->cls_k);
    E->cls_mode = E->cls_m[0] > 0 && E->cls_n[0] > 0 && E->cls_k[0] > 0;
  }
  // processing order of the numeric phase: column panels sized for the Infinity Cache, rows dealt to XCDs
  // size of B from the mean block sizes when the histograms are at hand (mixed sizes: the maxima overestimate it 2x on
  // BASELINE config 3, which doubled the number of panels and with it the compulsory re-reads of the A block-rows)
  double mean_k = E->max_k, mean_n = E->max_n;
  if (E->use_classes > 0 && E->max_k <= 32 && E->max_n <= 32 && E->min_k >= 1 && E->min_n >= 1) {
    double sk = 0, ck = 0, sn = 0, cn = 0;
    for (int sz = 1; sz <= 32; ++sz) {
      sn += (double)sz * E->cls_host_hist[33 + sz];
      cn += E->cls_host_hist[33 + sz];
      sk += (double)sz * E->cls_host_hist[66 + sz];
      ck += E->cls_host_hist[66 + sz];
    }
    if (ck > 0 && cn > 0) mean_k = sk / ck, mean_n = sn / cn;
  }
  const int64_t b_bytes_est = (int64_t)((double)b->nblks * mean_k * mean_n * (double)sizeof(double));
  int NP = (int)std::min<int64_t>((b_bytes_est + E->panel_bytes - 1) / E->panel_bytes, (int64_t)W);
  if (NP < 1) NP = 1;
  const int PW = (W + NP - 1) / NP;
  NP = (W + PW - 1) / PW;
  const int R = (nbr + 7) / 8;
  // rows walked together per XCD.  Measured on config 2 (DBCSR_AMD_MM_ROW_GROUP = 1/2/4/6/8: 22.7/22.8/23.5/25.2/26.3 ms):
  // the B reuse it buys (10 % fill: 14 % fewer B fetches at 4 rows) does not pay for the extra A rows in L2 -> default 1.
  int RG = E->row_group > 0 ? E->row_group : 1;
  RG = std::max(1, std::min(RG, R));
  const int NG = (R + RG - 1) / RG;
  const int nkeys = (E->cls_mode ? kNumClasses : 1) * 8 * NP * (E->cls_mode ? R : NG);
  if (E->order_cnt.ensure((size_t)nkeys + 1) || E->order_base.ensure((size_t)nkeys + 1)) return -1;
  if (E->cls_mode) {
    if (E->cls_row.ensure((size_t)nbr + 1) || E->cls_col.ensure((size_t)nbc + 1) || E->cls_col_bm.ensure((size_t)4 * W + 1) ||
        E->cls_lens.ensure(2 * kNumClasses + 1))
      return -1;
    hipLaunchKernelGGL(class_ids, grid_for(nbr), dim3(256), 0, st, a->row_blk_size, nbr, E->cls_m[0], E->cls_m[1], E->cls_m[2], E->cls_row.p);
    hipLaunchKernelGGL(class_ids, grid_for(nbc), dim3(256), 0, st, b->col_blk_size, nbc, E->cls_n[0], E->cls_n[1], E->cls_n[2], E->cls_col.p);
    hipLaunchKernelGGL(class_col_bitmaps, grid_for(W), dim3(256), 0, st, E->cls_col.p, nbc, W, E->cls_col_bm.p);
    hipLaunchKernelGGL(order_count_cls, grid_for(nkeys), dim3(256), 0, st, E->c_bm.p, E->cls_row.p, E->cls_col_bm.p, nbr, W, PW, NP, R,
                       E->order_cnt.p);
    if (exclusive_scan<int64_t>(E, E->order_cnt.p, nkeys, E->order_base.p, nullptr, false, st)) return -1;
    hipLaunchKernelGGL(order_len_cls, dim3(1), dim3(1), 0, st, E->order_base.p, c_nblks, NP, R, E->cls_lens.p);
    ACC_CHECK(hipMemcpyAsync(E->cls_host_lens, E->cls_lens.p, (2 * kNumClasses + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  } else {
  hipLaunchKernelGGL(order_count, grid_for(nkeys), dim3(256), 0, st, E->c_pre.p, E->row_nnz.p, nbr, W, PW, NP, NG, RG, E->order_cnt.p);
  if (exclusive_scan<int64_t>(E, E->order_cnt.p, nkeys, E->order_base.p, nullptr, false, st)) return -1;
  hipLaunchKernelGGL(order_len, dim3(1), dim3(1), 0, st, E->order_base.p, c_nblks, NP, NG, dsc + 7);
  }
  if (E->prod_cnt.ensure((size_t)c_nblks + 1) || E->blk_nze.ensure((size_t)c_nblks + 1) || E->prod_start.ensure((size_t)c_nblks + 1) ||
      E->c_blk_p_ws.ensure((size_t)c_nblks + 1))
    return -1;
  // 3. per C block: number of products, size; flop
  // one lane per (row, column) candidate unless C is extremely sparse (then one thread per bitmap word)
  const int nJ = (nbc + 63) / 64;
  E->grid_kernels = filtering || (((int64_t)nbr * nJ * 64 <= 256 * std::max<int64_t>(c_nblks, 1)) && !E->force_word_kernels);
  // sparse C (less than 60 % of the candidates are blocks) with enough block rows to fill the chip: product-driven kernels
  E->rows_kernels = E->force_symbolic == 3 || (E->force_symbolic == 0 && nbr >= 2048 && 10 * c_nblks < 6 * (int64_t)nbr * nbc);
  if (E->rows_kernels) {
    E->grid_kernels = false;
    ACC_CHECK(hipMemsetAsync(E->prod_cnt.p, 0, sizeof(int) * (size_t)c_nblks, st));
    hipLaunchKernelGGL(block_sizes_rows, grid_for((int64_t)nbr * W), dim3(256), 0, st, E->c_bm.p, E->c_pre.p, c_out_row_p, a->row_blk_size,
                       b->col_blk_size, nbr, W, E->blk_nze.p);
    hipLaunchKernelGGL(count_products_rows, grid_for((int64_t)nbr * 64), dim3(256), 0, st, a->row_p, a->col_i, a->row_blk_size, a->col_blk_size,
                       b->col_blk_size, b->row_p, b->col_i, E->c_bm.p, E->c_pre.p, c_out_row_p, nbr, W, E->prod_cnt.p, E->dev_scalars.p + 3,
                       E->filter);
  } else if (E->grid_kernels)
    hipLaunchKernelGGL(count_products_grid, grid_for((int64_t)nbr * nJ * 64), dim3(256), 0, st, a->row_p, a->col_i, a->row_blk_size,
                       a->col_blk_size, b->col_blk_size, E->b_bm.p, E->c_bm.p, E->c_pre.p, c_out_row_p, nbr, nbc, W, nJ,
                       E->prod_cnt.p, E->blk_nze.p, E->dev_scalars.p + 3, b->row_p, E->b_pre.p, E->filter);
  else
    hipLaunchKernelGGL(count_products, grid_for((int64_t)nbr * W), dim3(256), 0, st, a->row_p, a->col_i, a->row_blk_size,
                       a->col_blk_size, b->col_blk_size, E->b_bm.p, E->c_bm.p, E->c_pre.p, c_out_row_p, nbr, W, E->prod_cnt.p,
                       E->blk_nze.p, E->dev_scalars.p + 3);
  if (exclusive_scan<int64_t>(E, E->prod_cnt.p, c_nblks, E->prod_start.p, dsc + 2, false, st)) return -1;
  if (exclusive_scan<int64_t>(E, E->blk_nze.p, c_nblks, E->c_blk_p_ws.p, dsc + 1, false, st)) return -1;
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  E->order_len = E->host_scalars[7];
  if (E->cls_mode) {
    for (int c = 0; c < kNumClasses; ++c) {
      E->cls_len[c] = E->cls_host_lens[c];
      E->cls_off[c] = E->cls_host_lens[kNumClasses + c];
    }
    const int64_t total = E->cls_host_lens[2 * kNumClasses];
    E->order_len = total / 8;  // (only its product with 8 is used below: the size of order[])
    if (E->order.ensure((size_t)total + 64)) return -1;
    if (total > 0) {
      ACC_CHECK(hipMemsetAsync(E->order.p, 0xff, sizeof(int) * ((size_t)total + 64), st));
      hipLaunchKernelGGL(order_fill_cls, grid_for((int64_t)nbr * W), dim3(256), 0, st, E->c_bm.p, E->c_pre.p, c_out_row_p, E->cls_row.p,
                         E->cls_col_bm.p, E->order_base.p, E->cls_lens.p, nbr, W, PW, NP, R, E->order.p);
    }
  } else {
  if (E->order.ensure((size_t)(8 * E->order_len) + 64)) return -1;
  if (E->order_len > 0) {
    ACC_CHECK(hipMemsetAsync(E->order.p, 0xff, sizeof(int) * ((size_t)(8 * E->order_len) + 64), st));  // padding included
    hipLaunchKernelGGL(order_fill, grid_for((int64_t)nbr * W), dim3(256), 0, st, E->c_bm.p, E->c_pre.p, E->row_nnz.p, c_out_row_p,
                       E->order_base.p, nbr, W, PW, NP, NG, RG, E->order_len, E->order.p);
  }
  }
  counts->c_nblks = E->host_scalars[0];
  counts->c_nze = E->host_scalars[1];
  counts->nproducts = E->host_scalars[2];
  counts->flop = E->host_scalars[3];
  E->c_nblks = counts->c_nblks;
  E->nproducts = counts->nproducts;
  E->valid = true;
  ++E->plan_misses;
  if (!filtering) {
    if (plan_save(E, a, b, c_in, retain_sparsity ? 1 : 0, c_out_row_p, *counts, st)) return -1;
  } else {
    plan_invalidate(E);
  }
  return check(hipGetLastError(), "dbcsr_amd_mm_symbolic", __FILE__, __LINE__);
}

int dbcsr_amd_mm_numeric(void* handle, libsmm_acc_data_t datatype, double alpha, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b,
                         double beta, const dbcsr_amd_bcsr* c_in, dbcsr_amd_bcsr* c_out, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !E->valid || !a || !b || !c_in || !c_out) {
    fprintf(stderr, "dbcsr_amd_mm_numeric: no valid symbolic phase for this handle\n");
    return -1;
  }
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int nbr = E->nbr, W = E->W;
  const int64_t nblk = E->c_nblks;
  if (nblk == 0) return 0;
  // plan reuse: product lists, descriptors and launch order of the previous multiply stand; C's index is copied from the saved one
  const bool reuse = E->plan_hit && E->plan_numeric;
  if (!reuse) E->work_built = E->tile_built = false;
  if (E->entries.ensure((size_t)E->nproducts + 1) || E->descs.ensure((size_t)nblk + 1)) return -1;
  ACC_CHECK(hipEventRecord(E->ev[0], st));
  if (reuse) {
    ACC_CHECK(hipMemcpyAsync(c_out->col_i, E->plan_c_col_i.p, sizeof(int32_t) * (size_t)nblk, hipMemcpyDeviceToDevice, st));
    ACC_CHECK(hipMemcpyAsync(c_out->blk_p, E->plan_c_blk_p.p, sizeof(int64_t) * (size_t)nblk, hipMemcpyDeviceToDevice, st));
  } else if (E->rows_kernels) {
    if (E->tmp_i32.ensure((size_t)nblk + 1)) return -1;
    ACC_CHECK(hipMemsetAsync(E->tmp_i32.p, 0, sizeof(int) * (size_t)nblk, st));
    hipLaunchKernelGGL(fill_products_rows, grid_for((int64_t)nbr * 64), dim3(256), 0, st, a->row_p, a->col_i, a->blk_p, a->col_blk_size, b->row_p,
                       b->col_i, b->blk_p, E->c_bm.p, E->c_pre.p, c_out->row_p, E->prod_start.p, nbr, W, E->tmp_i32.p, E->entries.p, E->filter);
    hipLaunchKernelGGL(finish_descs_rows, grid_for((int64_t)nbr * W), dim3(256), 0, st, c_in->row_p, c_in->blk_p, c_out->row_blk_size,
                       c_out->col_blk_size, E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr,
                       E->have_cin ? E->cin_pre.p : (const int*)nullptr, E->c_bm.p, E->c_pre.p, c_out->row_p, E->c_blk_p_ws.p, E->prod_start.p,
                       E->prod_cnt.p, nbr, W, c_out->col_i, c_out->blk_p, E->descs.p);
  } else if (E->grid_kernels) {
    const int nbc = b->nblkcols, nJ = (nbc + 63) / 64;
    hipLaunchKernelGGL(fill_products_grid, grid_for((int64_t)nbr * nJ * 64), dim3(256), 0, st, a->row_p, a->col_i, a->blk_p, b->row_p,
                       b->blk_p, c_in->row_p, c_in->blk_p, a->row_blk_size, a->col_blk_size, b->col_blk_size, E->b_bm.p, E->b_pre.p,
                       E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr, E->have_cin ? E->cin_pre.p : (const int*)nullptr,
                       E->c_bm.p, E->c_pre.p, c_out->row_p, E->prod_start.p, E->c_blk_p_ws.p, nbr, nbc, W, nJ, c_out->col_i,
                       c_out->blk_p, E->descs.p, E->entries.p, E->filter);
  } else {
    hipLaunchKernelGGL(fill_products, grid_for((int64_t)nbr * W), dim3(256), 0, st, a->row_p, a->col_i, a->blk_p, b->row_p, b->blk_p,
                       c_in->row_p, c_in->blk_p, a->row_blk_size, a->col_blk_size, b->col_blk_size, E->b_bm.p, E->b_pre.p,
                       E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr, E->have_cin ? E->cin_pre.p : (const int*)nullptr,
                       E->c_bm.p, E->c_pre.p, c_out->row_p, E->prod_start.p, E->c_blk_p_ws.p, nbr, W, c_out->col_i, c_out->blk_p,
                       E->descs.p, E->entries.p);
  }
  const unsigned nwg = (unsigned)((nblk + 3) / 4);
  // waves per workgroup of the one-wave-per-C-block kernels.  A workgroup's LDS is released when its LAST wave ends: with short,
  // uneven product lists one wave per workgroup keeps more wave slots busy (config 3: kernel 8.93 -> 7.51 ms, generic LDS kernel
  // 12.3 -> 9.2, config 2: -5 %, config 4: -6 %); with long lists four waves per workgroup are faster (config 5, 164 products per
  // block: 2.03 s against 2.27 s)
  const int ww = E->wg_waves > 0 ? E->wg_waves : (E->nproducts <= 32 * nblk ? 1 : 4);
  // in-place accumulation (Cannon ticks after the first): C blocks without products in this call are left untouched
  const int skip_empty = (c_out->data == c_in->data && E->retain && beta == 1.0) ? 1 : 0;
  // launch-order work records for the exact-size fp64 kernels (one wave per C block): descriptor + first product in one read
  const Work* hot_work = nullptr;
  {
    const bool small64 = datatype == dbcsr_type_real_8 && E->use_lds && E->max_m <= 32 && E->max_k <= 32 && E->max_n <= 32 && E->min_m >= 1 &&
                         E->min_k >= 1 && E->min_n >= 1 && !(E->use_tiny && E->max_m <= 4 && E->max_n <= 4);
    // (the ahead-of-time exact-size kernel reads nothing else; the class kernels keep the order[] -> descs[] path for DBCSR_AMD_MM_WORK=0)
    const bool exact = E->cls_mode ? (E->class_g == 1 && E->use_work)
                                   : (E->use_hot && E->use_pipe != 1 && E->hot_m > 0 && E->dma_stages == 0 && E->hot_m == E->hot_n && E->hot_m == E->hot_k);
    const int64_t npos = 8 * E->order_len;
    if (small64 && exact && npos > 0) {
      if (!(reuse && E->work_built)) {
        if (E->work.ensure((size_t)npos + 1)) return -1;
        hipLaunchKernelGGL(build_work, grid_for(npos), dim3(256), 0, st, E->order.p, npos, E->descs.p, nblk, E->entries.p, E->work.p);
        E->work_built = true;
      }
      hot_work = E->work.p;
    }
  }
  // a filtered multiply ends with the block filter on C's norms: the exact-size kernel leaves them behind (dbcsr_amd_bcsr_filter_count
  // then skips its pass over C)
  double* epi_norms = nullptr;
  E->norms_data = nullptr;
  if ((hot_work || (E->cls_mode && E->class_g == 1)) && datatype == dbcsr_type_real_8 && E->filter.a_norms && !skip_empty && !E->retain) {
    if (E->norms64.ensure((size_t)nblk + 1)) return -1;
    epi_norms = E->norms64.p;
    if (E->dbg & 8) epi_norms = nullptr;  // (profiling epilogue of the exact-size kernel: it leaves no norms, the filter then computes them)
  }
  ACC_CHECK(hipEventRecord(E->ev[1], st));
  if (datatype == dbcsr_type_real_8) {
    // LDS path: blocks of at most 32 x 32 (any smaller size: the staging loads are bounds-checked buffer loads)
    const bool small = E->max_m <= 32 && E->max_k <= 32 && E->max_n <= 32 && E->min_m >= 1 && E->min_k >= 1 && E->min_n >= 1;
    if (E->use_tiny && E->max_m <= 4 && E->max_n <= 4 && E->min_m >= 1 && E->min_n >= 1 && E->min_k >= 1) {
      // four C blocks per wave, one per MFMA sub-block; order[] is padded to a multiple of 4 per XCD stream, so a wave's
      // four positions never straddle two streams only if the stream length is a multiple of 16: the tail positions hold -1
      const unsigned nwg_t = (unsigned)((8 * E->order_len + 15) / 16);
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_tiny");
      hipLaunchKernelGGL(mm_numeric_f64_tiny, dim3(nwg_t), dim3(256), 0, st, E->descs.p, nblk, E->entries.p,
                         static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),
                         static_cast<const double*>(c_in->data), alpha, beta, skip_empty, E->order.p);
    } else if (small && E->use_lds && E->cls_mode) {
      // one launch per (m, n) class on its segment of order[]: the run-time compiled exact-size kernel of the class
      // (mm_exact.h, mm_jit.hip), the generic LDS kernel for class 9 (other sizes) and for classes hiprtc could not serve
      const int g_lds_a = (E->max_m * ((E->max_k + 3) & ~3) + 1) & ~1, g_lds_b = ((E->max_k * E->max_n + 127) / 128) * 128;
      const int g_lds_wave = g_lds_a + g_lds_b;
      const int g_maxt = (std::max(E->max_m, E->max_n) + 7) / 8;
      const int dbgv = E->dbg | (skip_empty ? 32 : 0);
      int njit = 0, ngen = 0, jit_mask = 0;
      for (int c = 0; c < kNumClasses; ++c) {
        if (E->cls_len[c] == 0) continue;
        const int* ord = E->order.p + E->cls_off[c];
        const unsigned nwg_c = (unsigned)(8 * E->cls_len[c] / 4);
        ClassKernel ck;
        const int cm = c < 9 ? E->cls_m[c / 3] : 0, cn = c < 9 ? E->cls_n[c % 3] : 0;
        if (c < 9 && cm > 0 && cn > 0 && jit_class_kernel(cm, cn, E->cls_k[0], E->cls_k[1], E->cls_k[2], E->class_g, &ck) == 0) {
          const Desc* p_descs = E->descs.p;
          long p_nblk = (long)nblk;
          const Entry* p_entries = E->entries.p;
          const double* p_a = static_cast<const double*>(a->data);
          const double* p_b = static_cast<const double*>(b->data);
          double* p_c = static_cast<double*>(c_out->data);
          const double* p_ci = static_cast<const double*>(c_in->data);
          double p_alpha = alpha, p_beta = beta;
          int p_skip = skip_empty;
          const Work* p_work = hot_work ? hot_work + E->cls_off[c] : nullptr;
          double* p_norms = epi_norms;
          void* args[] = {&p_descs, &p_nblk, &p_entries, &p_a, &p_b, &p_c, &p_ci, &p_alpha, &p_beta, &p_skip, &ord, &p_work, &p_norms};
          jit_mask |= 1 << c;
          const unsigned cw = E->class_g == 1 ? (unsigned)ww : 4u;  // waves per workgroup (the G-block stream body keeps 4)
          ACC_CHECK(hipModuleLaunchKernel(ck.fn, (unsigned)(8 * E->cls_len[c]) / cw / (unsigned)E->class_g, 1, 1, 64 * cw, 1, 1,
                                          (unsigned)(cw * ck.wave_lds), st, args, nullptr));
          ++njit;
        } else {
          const size_t lb = (size_t)ww * g_lds_wave * sizeof(double);
#define DBCSR_LAUNCH_G(T_)                                                                                                         \
  hipLaunchKernelGGL(mm_numeric_f64_lds<T_>, dim3(nwg_c * 4u / (unsigned)ww), dim3(64 * ww), lb, st, E->descs.p, nblk, E->entries.p, \
                     static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),   \
                     static_cast<const double*>(c_in->data), alpha, beta, g_lds_a, g_lds_wave, dbgv, ord)
          switch (g_maxt) {
            case 1: DBCSR_LAUNCH_G(1); break;
            case 2: DBCSR_LAUNCH_G(2); break;
            case 3: DBCSR_LAUNCH_G(3); break;
            default: DBCSR_LAUNCH_G(4); break;
          }
#undef DBCSR_LAUNCH_G
          ++ngen;
        }
      }
      if (epi_norms) {  // the blocks the generic kernel handled did not leave their norm
        ClassSet cs;
        for (int q = 0; q < 3; ++q) cs.m[q] = E->cls_m[q], cs.n[q] = E->cls_n[q];
        cs.jit_mask = jit_mask;
        hipLaunchKernelGGL(block_norms_unserved_classes, grid_for(nblk * 64), dim3(256), 0, st, E->descs.p, nblk,
                           static_cast<const double*>(c_out->data), cs, epi_norms);
        E->norms_data = c_out->data;
        E->norms_nblks = nblk;
      }
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_class[%d jit + %d generic launches; m {%d,%d,%d} n {%d,%d,%d} k {%d,%d,%d}]", njit,
               ngen, E->cls_m[0], E->cls_m[1], E->cls_m[2], E->cls_n[0], E->cls_n[1], E->cls_n[2], E->cls_k[0], E->cls_k[1], E->cls_k[2]);
    } else if (small && E->use_lds) {
      // per-wave LDS slice.  Staging writes whole 1 KiB chunks (128 doubles), A's chunks first, then B's: the B part may
      // start right after A's (zero-padded) block -- the tail of A's last chunk is simply overwritten by B's first chunk
      // (one wave, in-order LDS queue) -- and only B's part is rounded up to whole chunks.  For 23x23 blocks this is
      // 9.5 KB per wave instead of 10 KB, which is what lets a 4th workgroup (16 waves) fit the CU's 160 KB.
      const int lds_a = (E->max_m * ((E->max_k + 3) & ~3) + 1) & ~1, lds_b = ((E->max_k * E->max_n + 127) / 128) * 128;
      const int lds_wave = lds_a + lds_b;
      const int maxt = (std::max(E->max_m, E->max_n) + 7) / 8;
      const size_t lds_bytes = (size_t)4 * lds_wave * sizeof(double) + (size_t)E->lds_pad;
      const unsigned nwg_o = (unsigned)(8 * E->order_len / 4);
#define DBCSR_LAUNCH(T_)                                                                                                        \
  hipLaunchKernelGGL(mm_numeric_f64_lds<T_>, dim3(nwg_o * 4u / (unsigned)ww), dim3(64 * ww),                     \
                     (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad, st, E->descs.p, nblk, E->entries.p,     \
                     static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data), \
                     static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, E->dbg | (skip_empty ? 32 : 0), E->order.p)
      // XCD-wide C tiles (mm_tile.h): one dominant cube size the tile kernel is built for, a C dense enough that sub-tiles of
      // 3 x 3 blocks have long product lists, no on-the-fly filter, no in-place accumulation, no symmetric product
      int tile_rc = 1;
      if (E->use_tile > 0 && hot_work && E->use_hot && E->use_pipe != 1 && E->dma_stages == 0 && E->hot_m == 23 && E->hot_n == 23 && E->hot_k == 23 &&
          !E->filter.a_norms && !skip_empty && !E->canonical_c && !epi_norms && !(E->dbg & ~32) &&
          (E->use_tile > 1 || (E->nproducts >= 8 * nblk && nblk >= 200000)))
        tile_rc = run_tile_f64<23>(E, st, a, b, c_in, c_out, alpha, beta);
      if (tile_rc < 0) return -1;
      // measured: the pipelined kernel wins when C blocks have few products (config 3: 3.7 per block, 10.4 vs 11.8 ms) and
      // loses when they have many (config 2: 14.4 per block, 32 vs 22 ms)
      if (tile_rc == 0) {
        // the tile kernel computed the C blocks of the dominant size (products with inner blocks of another size included);
        // this launch: the exact-size kernel over the blocks of the other sizes only
        launch_hot_f64(E->hot_m, E->hot_n, E->hot_k, dim3((unsigned)(8 * E->order_len / ww)), (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad,
                       st, E->descs.p, nblk, E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                       static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, 64, E->order.p,
                       hot_work, ww, nullptr, 0);
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_tile<%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k);
      } else if (E->use_hot && E->use_pipe != 1 && E->hot_m > 0 && E->dma_stages > 0 &&
          launch_dma_f64(E->dma_stages, E->hot_m, E->hot_n, E->hot_k, (unsigned)(8 * E->order_len), st, E->descs.p, nblk, E->entries.p,
                         static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),
                         static_cast<const double*>(c_in->data), alpha, beta, skip_empty, E->order.p)) {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_dma<%d,%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k, E->dma_stages);
      } else if (E->use_hot && E->use_pipe != 1 && E->hot_m > 0 &&
          launch_hot_f64(E->hot_m, E->hot_n, E->hot_k, dim3((unsigned)(8 * E->order_len / ww)),
                         (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad, st, E->descs.p, nblk, E->entries.p,
                         static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),
                         static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, E->dbg | (skip_empty ? 32 : 0), E->order.p,
                         hot_work, ww, epi_norms, (E->dbg & ~32) ? 1 : E->hot_variant)) {
        // launched: C blocks of the dominant size take the exact-size path, the others the generic one
        if (epi_norms) {  // blocks of another size (tail block row / column) did not leave their norm: a pass over those only
          hipLaunchKernelGGL(block_norms_other_sizes, grid_for(nblk * 64), dim3(256), 0, st, E->descs.p, nblk, static_cast<const double*>(c_out->data),
                             E->hot_m, E->hot_n, epi_norms);
          E->norms_data = c_out->data;
          E->norms_nblks = nblk;
        }
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_hot<%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k);
      } else {
      const bool pipe = E->use_pipe == 1 || (E->use_pipe < 0 && E->nproducts < 6 * nblk && E->nproducts > nblk + nblk / 2);
      if (pipe) {
        const int64_t npos = 8 * E->order_len;
        const int G = E->pipe_g;
        const unsigned nwg_p = (unsigned)((npos + 4 * (int64_t)G - 1) / (4 * (int64_t)G));
#define DBCSR_LAUNCH_P(T_)                                                                                                       \
  hipLaunchKernelGGL(mm_numeric_f64_pipe<T_>, dim3(nwg_p), dim3(256), lds_bytes, st, E->descs.p, nblk, E->entries.p,             \
                     static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data), \
                     static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, skip_empty, E->order.p, npos, G)
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_pipe<%d>", maxt > 4 ? 4 : maxt);
        switch (maxt) {
          case 1: DBCSR_LAUNCH_P(1); break;
          case 2: DBCSR_LAUNCH_P(2); break;
          case 3: DBCSR_LAUNCH_P(3); break;
          default: DBCSR_LAUNCH_P(4); break;
        }
#undef DBCSR_LAUNCH_P
      } else {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_lds<%d>", maxt > 4 ? 4 : maxt);
        switch (maxt) {
          case 1: DBCSR_LAUNCH(1); break;
          case 2: DBCSR_LAUNCH(2); break;
          case 3: DBCSR_LAUNCH(3); break;
          default: DBCSR_LAUNCH(4); break;
        }
      }
      }
#undef DBCSR_LAUNCH
    } else {
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64");
      hipLaunchKernelGGL(mm_numeric_f64, dim3(nwg), dim3(256), 0, st, E->descs.p, nblk, E->entries.p,
                         static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                         static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, skip_empty);
    }
  } else {
    const bool small32 = E->max_m <= 32 && E->max_k <= 32 && E->max_n <= 32 && E->min_m >= 1 && E->min_k >= 1 && E->min_n >= 1;
    if (small32 && E->use_lds && E->cls_mode) {
      for (int c = 0; c < kNumClasses; ++c) {
        if (E->cls_len[c] == 0) continue;
        hipLaunchKernelGGL(mm_numeric_f32_lds, dim3((unsigned)(8 * E->cls_len[c] / ww)), dim3(64 * ww), f32_lds_bytes(ww), st, E->descs.p, nblk, E->entries.p,
                           static_cast<const float*>(a->data), static_cast<const float*>(b->data), static_cast<float*>(c_out->data),
                           static_cast<const float*>(c_in->data), (float)alpha, (float)beta, skip_empty, E->order.p + E->cls_off[c]);
      }
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32_lds[per class segment]");
    } else if (small32 && E->use_lds) {
      const unsigned nwg_o = (unsigned)(8 * E->order_len / ww);
      if (E->use_hot && E->hot_m > 0 &&
          launch_hot_f32(E->hot_m, E->hot_n, E->hot_k, dim3(nwg_o), ww, st, E->descs.p, nblk, E->entries.p, static_cast<const float*>(a->data),
                         static_cast<const float*>(b->data), static_cast<float*>(c_out->data), static_cast<const float*>(c_in->data),
                         (float)alpha, (float)beta, skip_empty, E->order.p)) {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32_hot<%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k);
      } else {
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32_lds");
      hipLaunchKernelGGL(mm_numeric_f32_lds, dim3(nwg_o), dim3(64 * ww), f32_lds_bytes(ww), st, E->descs.p, nblk, E->entries.p,
                         static_cast<const float*>(a->data), static_cast<const float*>(b->data), static_cast<float*>(c_out->data),
                         static_cast<const float*>(c_in->data), (float)alpha, (float)beta, skip_empty, E->order.p);
      }
    } else {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32");
        hipLaunchKernelGGL(mm_numeric_f32, dim3(nwg), dim3(256), 0, st, E->descs.p, nblk, E->entries.p,
                       static_cast<const float*>(a->data), static_cast<const float*>(b->data), static_cast<float*>(c_out->data),
                       static_cast<const float*>(c_in->data), (float)alpha, (float)beta, skip_empty);
    }
  }
  ACC_CHECK(hipEventRecord(E->ev[2], st));
  E->timed = true;
  c_out->nblks = nblk;
  if (E->plan_saved && !E->plan_numeric) {  // first numeric phase of a saved plan: keep C's index for the multiplies that reuse it
    if (E->plan_c_col_i.ensure((size_t)nblk + 1) || E->plan_c_blk_p.ensure((size_t)nblk + 1)) return -1;
    ACC_CHECK(hipMemcpyAsync(E->plan_c_col_i.p, c_out->col_i, sizeof(int32_t) * (size_t)nblk, hipMemcpyDeviceToDevice, st));
    ACC_CHECK(hipMemcpyAsync(E->plan_c_blk_p.p, c_out->blk_p, sizeof(int64_t) * (size_t)nblk, hipMemcpyDeviceToDevice, st));
    E->plan_numeric = true;
  }
  return check(hipGetLastError(), "dbcsr_amd_mm_numeric", __FILE__, __LINE__);
}


int dbcsr_amd_mm_init_c(void* handle, libsmm_acc_data_t datatype, double beta, const dbcsr_amd_bcsr* c_in, dbcsr_amd_bcsr* c_out,
                        void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !E->valid || !c_in || !c_out) {
    fprintf(stderr, "dbcsr_amd_mm_init_c: no valid symbolic phase for this handle\n");
    return -1;
  }
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int nbr = E->nbr, W = E->W;
  const int64_t nblk = E->c_nblks;
  c_out->nblks = nblk;
  if (nblk == 0) return 0;
  if (E->descs.ensure((size_t)nblk + 1)) return -1;
  hipLaunchKernelGGL(emit_index, grid_for((int64_t)nbr * W), dim3(256), 0, st, c_in->row_p, c_in->blk_p, c_out->row_blk_size,
                     c_out->col_blk_size, E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr,
                     E->have_cin ? E->cin_pre.p : (const int*)nullptr, E->c_bm.p, E->c_pre.p, c_out->row_p, E->c_blk_p_ws.p, nbr, W,
                     c_out->col_i, c_out->blk_p, E->descs.p);
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((init_c_blocks<double>), grid_for(nblk * 64), dim3(256), 0, st, E->descs.p, nblk,
                       static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), beta);
  else
    hipLaunchKernelGGL((init_c_blocks<float>), grid_for(nblk * 64), dim3(256), 0, st, E->descs.p, nblk,
                       static_cast<float*>(c_out->data), static_cast<const float*>(c_in->data), (float)beta);
  return check(hipGetLastError(), "dbcsr_amd_mm_init_c", __FILE__, __LINE__);
}


static int element_offsets(Engine* E, const int* sizes, int n, DevBuf<int64_t>& off, hipStream_t st) {
  if (off.ensure((size_t)n + 1)) return -1;
  return exclusive_scan<int64_t>(E, sizes, n, off.p, nullptr, false, st);
}

static Window make_window(const dbcsr_amd_bcsr* m, int64_t row_lo, int64_t row_hi, int64_t col_lo, int64_t col_hi) {
  (void)m;
  const int64_t big = 0x7fffffff;
  Window w;
  w.r0 = (int)(row_lo < 0 ? 0 : row_lo);
  w.r1 = (int)(row_hi < 0 || row_hi > big ? big : row_hi);
  w.c0 = (int)(col_lo < 0 ? 0 : col_lo);
  w.c1 = (int)(col_hi < 0 || col_hi > big ? big : col_hi);
  return w;
}

int dbcsr_amd_bcsr_crop_count(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, int64_t row_lo, int64_t row_hi,
                              int64_t col_lo, int64_t col_hi, int32_t* new_row_p, int64_t* new_nblks, int64_t* new_nze, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !m || !new_row_p || !new_nblks || !new_nze) return -1;
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int nbr = m->nblkrows;
  const int64_t nb = m->nblks;
  E->valid = false;  // shares workspace with the symbolic phase
  E->flt_nblks = nb;
  E->crop_win = make_window(m, row_lo, row_hi, col_lo, col_hi);
  E->crop_pending = true;
  if (E->keep.ensure((size_t)nb + 1) || E->blk_nze.ensure((size_t)nb + 1) || E->row_nnz.ensure((size_t)nbr + 1) ||
      E->prod_start.ensure((size_t)nb + 1) || E->c_blk_p_ws.ensure((size_t)nb + 1) || E->dev_scalars.ensure(16))
    return -1;
  int64_t* dsc = reinterpret_cast<int64_t*>(E->dev_scalars.p);
  ACC_CHECK(hipMemsetAsync(dsc, 0, 16 * sizeof(int64_t), st));
  if (element_offsets(E, m->row_blk_size, nbr, E->off_a, st)) return -1;
  if (element_offsets(E, m->col_blk_size, m->nblkcols, E->off_b, st)) return -1;
  if (nbr > 0 && nb > 0)
    hipLaunchKernelGGL(crop_flags, grid_for((int64_t)nbr * 64), dim3(256), 0, st, m->row_p, m->col_i, m->row_blk_size, m->col_blk_size,
                       E->off_a.p, E->off_b.p, nbr, E->crop_win, E->keep.p, E->blk_nze.p, E->row_nnz.p);
  else if (nbr > 0)
    ACC_CHECK(hipMemsetAsync(E->row_nnz.p, 0, sizeof(int) * (size_t)nbr, st));
  if (exclusive_scan<int32_t>(E, E->row_nnz.p, nbr, new_row_p, dsc + 0, true, st)) return -1;
  if (exclusive_scan<int64_t>(E, E->keep.p, nb, E->prod_start.p, nullptr, false, st)) return -1;
  if (exclusive_scan<int64_t>(E, E->blk_nze.p, nb, E->c_blk_p_ws.p, dsc + 1, false, st)) return -1;
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  *new_nblks = E->host_scalars[0];
  *new_nze = E->host_scalars[1];
  return check(hipGetLastError(), "dbcsr_amd_bcsr_crop_count", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_crop_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, dbcsr_amd_bcsr* dst, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !src || !dst || !E->crop_pending || E->flt_nblks != src->nblks) return -1;
  E->crop_pending = false;
  hipStream_t st = stream_of(stream);
  const int nbr = src->nblkrows;
  if (nbr == 0 || src->nblks == 0) return 0;
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((crop_compact<double>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const double*>(src->data), src->row_blk_size, src->col_blk_size, E->off_a.p, E->off_b.p, nbr,
                       E->crop_win, E->keep.p, E->prod_start.p, E->c_blk_p_ws.p, dst->col_i, dst->blk_p, static_cast<double*>(dst->data));
  else if (datatype == dbcsr_type_real_4)
    hipLaunchKernelGGL((crop_compact<float>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const float*>(src->data), src->row_blk_size, src->col_blk_size, E->off_a.p, E->off_b.p, nbr,
                       E->crop_win, E->keep.p, E->prod_start.p, E->c_blk_p_ws.p, dst->col_i, dst->blk_p, static_cast<float*>(dst->data));
  else
    return -10;
  return check(hipGetLastError(), "dbcsr_amd_bcsr_crop_apply", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_scale_window(void* handle, libsmm_acc_data_t datatype, dbcsr_amd_bcsr* m, double beta, int64_t row_lo, int64_t row_hi,
                                int64_t col_lo, int64_t col_hi, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !m) return -1;
  hipStream_t st = stream_of(stream);
  const int nbr = m->nblkrows;
  if (nbr == 0 || m->nblks == 0) return 0;
  const Window w = make_window(m, row_lo, row_hi, col_lo, col_hi);
  if (element_offsets(E, m->row_blk_size, nbr, E->off_a, st)) return -1;
  if (element_offsets(E, m->col_blk_size, m->nblkcols, E->off_b, st)) return -1;
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((scale_window<double>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<double*>(m->data), m->row_blk_size, m->col_blk_size, E->off_a.p, E->off_b.p, nbr, w, beta);
  else if (datatype == dbcsr_type_real_4)
    hipLaunchKernelGGL((scale_window<float>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<float*>(m->data), m->row_blk_size, m->col_blk_size, E->off_a.p, E->off_b.p, nbr, w, (float)beta);
  else
    return -10;
  return check(hipGetLastError(), "dbcsr_amd_bcsr_scale_window", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_filter_count(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, double eps, int32_t* new_row_p,
                                int64_t* new_nblks, int64_t* new_nze, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !m || !new_row_p || !new_nblks || !new_nze) return -1;
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int nbr = m->nblkrows;
  const int64_t nb = m->nblks;
  E->valid = false;  // shares workspace with the symbolic phase
  E->flt_nblks = nb;
  if (E->norms64.ensure((size_t)nb + 1) || E->keep.ensure((size_t)nb + 1) || E->blk_nze.ensure((size_t)nb + 1) ||
      E->row_nnz.ensure((size_t)nbr + 1) || E->prod_start.ensure((size_t)nb + 1) || E->c_blk_p_ws.ensure((size_t)nb + 1) ||
      E->dev_scalars.ensure(16))
    return -1;
  int64_t* dsc = reinterpret_cast<int64_t*>(E->dev_scalars.p);
  ACC_CHECK(hipMemsetAsync(dsc, 0, 16 * sizeof(int64_t), st));
  const bool have_norms = E->norms_data != nullptr && E->norms_data == m->data && E->norms_nblks == nb && datatype == dbcsr_type_real_8;
  E->norms_data = nullptr;
  if (nbr > 0 && nb > 0) {
    const int sm = row_split(nbr, nb);
    if (have_norms) {
      // left behind by the numeric kernel of the multiply that produced m
    } else if (datatype == dbcsr_type_real_8)
      hipLaunchKernelGGL((bcsr_block_norms<double>), grid_for((int64_t)nbr * sm * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                         static_cast<const double*>(m->data), m->row_blk_size, m->col_blk_size, nbr, sm, 1.0, (float*)nullptr, E->norms64.p);
    else
      hipLaunchKernelGGL((bcsr_block_norms<float>), grid_for((int64_t)nbr * sm * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                         static_cast<const float*>(m->data), m->row_blk_size, m->col_blk_size, nbr, sm, 1.0, (float*)nullptr, E->norms64.p);
    hipLaunchKernelGGL(filter_flags, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->norms64.p, nb, m->row_p, m->col_i, m->row_blk_size,
                       m->col_blk_size, nbr, eps * eps, E->keep.p, E->blk_nze.p, E->row_nnz.p);
  }
  if (exclusive_scan<int32_t>(E, E->row_nnz.p, nbr, new_row_p, dsc + 0, true, st)) return -1;
  if (exclusive_scan<int64_t>(E, E->keep.p, nb, E->prod_start.p, nullptr, false, st)) return -1;     // new index of each kept block
  if (exclusive_scan<int64_t>(E, E->blk_nze.p, nb, E->c_blk_p_ws.p, dsc + 1, false, st)) return -1;  // new data offset
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  *new_nblks = E->host_scalars[0];
  *new_nze = E->host_scalars[1];
  return check(hipGetLastError(), "dbcsr_amd_bcsr_filter_count", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_filter_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, dbcsr_amd_bcsr* dst, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !src || !dst || E->flt_nblks != src->nblks) return -1;
  hipStream_t st = stream_of(stream);
  const int nbr = src->nblkrows;
  if (nbr == 0 || src->nblks == 0) return 0;
  const int sc = row_split(nbr, src->nblks);
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((filter_compact<double>), grid_for((int64_t)nbr * sc * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const double*>(src->data), src->row_blk_size, src->col_blk_size, nbr, sc, E->keep.p, E->prod_start.p,
                       E->c_blk_p_ws.p, dst->col_i, dst->blk_p, static_cast<double*>(dst->data));
  else if (datatype == dbcsr_type_real_4)
    hipLaunchKernelGGL((filter_compact<float>), grid_for((int64_t)nbr * sc * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const float*>(src->data), src->row_blk_size, src->col_blk_size, nbr, sc, E->keep.p, E->prod_start.p,
                       E->c_blk_p_ws.p, dst->col_i, dst->blk_p, static_cast<float*>(dst->data));
  else
    return -10;
  return check(hipGetLastError(), "dbcsr_amd_bcsr_filter_apply", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_checksum(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, double* out2, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !m || !out2) return -1;
  hipStream_t st = stream_of(stream);
  const int nbr = m->nblkrows;
  out2[0] = out2[1] = 0.0;
  if (nbr == 0 || m->nblks == 0) return 0;
  if (E->row_sums.ensure((size_t)2 * nbr + 2)) return -1;
  if (element_offsets(E, m->row_blk_size, nbr, E->off_a, st)) return -1;
  if (element_offsets(E, m->col_blk_size, m->nblkcols, E->off_b, st)) return -1;
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((checksum_blocks<double>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<const double*>(m->data), m->row_blk_size, m->col_blk_size, E->off_a.p, E->off_b.p, nbr,
                       E->row_sums.p);
  else if (datatype == dbcsr_type_real_4)
    hipLaunchKernelGGL((checksum_blocks<float>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<const float*>(m->data), m->row_blk_size, m->col_blk_size, E->off_a.p, E->off_b.p, nbr,
                       E->row_sums.p);
  else
    return -10;
  hipLaunchKernelGGL(checksum_final, dim3(1), dim3(256), 0, st, E->row_sums.p, nbr, E->row_sums.p + 2 * (size_t)nbr);
  ACC_CHECK(hipMemcpyAsync(E->host_scalars + 4, E->row_sums.p + 2 * (size_t)nbr, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  memcpy(out2, E->host_scalars + 4, 2 * sizeof(double));
  return 0;
}

int dbcsr_amd_bcsr_fill_random(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, int counter, void* stream) {
  return dbcsr_amd_bcsr_fill_random_dist(handle, datatype, m, counter, nullptr, nullptr, m ? m->nblkrows : 0, stream);
}

int dbcsr_amd_bcsr_fill_random_dist(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, int counter,
                                    const int32_t* row_gid, const int32_t* col_gid, int32_t nblkrows_global, void* stream) {
  if (!handle || !m) return -1;
  hipStream_t st = stream_of(stream);
  if (m->nblks == 0) return 0;
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL(fill_random_f64, grid_for((int64_t)m->nblkrows * 64), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<double*>(m->data), m->row_blk_size, m->col_blk_size, m->nblkrows, m->nblkcols, counter, row_gid, col_gid,
                       nblkrows_global);
  else if (datatype == dbcsr_type_real_4)
    hipLaunchKernelGGL(fill_random_f32, grid_for(m->nblks), dim3(256), 0, st, m->row_p, m->col_i, m->blk_p,
                       static_cast<float*>(m->data), m->row_blk_size, m->col_blk_size, m->nblkrows, m->nblkcols, counter, row_gid, col_gid,
                       nblkrows_global);
  else
    return -10;
  return check(hipGetLastError(), "dbcsr_amd_bcsr_fill_random", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_transpose(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, dbcsr_amd_bcsr* dst, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !src || !dst) return -1;
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int s_nbr = src->nblkrows, t_nbr = src->nblkcols;
  const int Wt = (s_nbr + 31) / 32;
  E->valid = false;  // shares workspace with the symbolic phase
  if (E->c_bm.ensure((size_t)t_nbr * Wt + 1) || E->c_pre.ensure((size_t)t_nbr * Wt + 1) || E->row_nnz.ensure((size_t)t_nbr + 1) ||
      E->blk_nze.ensure((size_t)src->nblks + 1) || E->c_blk_p_ws.ensure((size_t)src->nblks + 1))
    return -1;
  if (t_nbr == 0) return 0;
  ACC_CHECK(hipMemsetAsync(E->c_bm.p, 0, sizeof(uint32_t) * (size_t)t_nbr * Wt, st));
  if (s_nbr > 0) hipLaunchKernelGGL(transpose_mark, grid_for((int64_t)s_nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, s_nbr, Wt, E->c_bm.p);
  hipLaunchKernelGGL(row_prefix, grid_for((int64_t)t_nbr * 64), dim3(256), 0, st, E->c_bm.p, t_nbr, Wt, E->c_pre.p, E->row_nnz.p);
  if (exclusive_scan<int32_t>(E, E->row_nnz.p, t_nbr, dst->row_p, nullptr, true, st)) return -1;
  if (src->nblks > 0) {
    hipLaunchKernelGGL(transpose_sizes, grid_for((int64_t)t_nbr * Wt), dim3(256), 0, st, E->c_bm.p, E->c_pre.p, dst->row_p,
                       src->row_blk_size, src->col_blk_size, t_nbr, Wt, E->blk_nze.p);
    if (exclusive_scan<int64_t>(E, E->blk_nze.p, src->nblks, E->c_blk_p_ws.p, nullptr, false, st)) return -1;
    if (datatype == dbcsr_type_real_8)
      hipLaunchKernelGGL((transpose_fill<double>), grid_for((int64_t)s_nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                         static_cast<const double*>(src->data), src->row_blk_size, src->col_blk_size, E->c_bm.p, E->c_pre.p, dst->row_p,
                         E->c_blk_p_ws.p, s_nbr, Wt, dst->col_i, dst->blk_p, static_cast<double*>(dst->data));
    else
      hipLaunchKernelGGL((transpose_fill<float>), grid_for((int64_t)s_nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                         static_cast<const float*>(src->data), src->row_blk_size, src->col_blk_size, E->c_bm.p, E->c_pre.p, dst->row_p,
                         E->c_blk_p_ws.p, s_nbr, Wt, dst->col_i, dst->blk_p, static_cast<float*>(dst->data));
  }
  dst->nblks = src->nblks;
  return check(hipGetLastError(), "dbcsr_amd_bcsr_transpose", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_desymmetrize_count(void* handle, const dbcsr_amd_bcsr* src, int32_t* dst_row_p, int64_t* nblks, int64_t* nze, void* stream) {
  return dbcsr_amd_bcsr_twin_count(handle, src, 0, dst_row_p, nblks, nze, stream);
}

int dbcsr_amd_bcsr_desymmetrize_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, int antisymmetric, dbcsr_amd_bcsr* dst,
                                      void* stream) {
  return dbcsr_amd_bcsr_twin_apply(handle, datatype, src, 0, antisymmetric, dst, stream);
}

int dbcsr_amd_mm_set_canonical_product(void* handle, int on) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E) return -1;
  E->canonical_c = on ? 1 : 0;
  return 0;
}

int dbcsr_amd_bcsr_twin_count(void* handle, const dbcsr_amd_bcsr* src, int mode, int32_t* dst_row_p, int64_t* nblks, int64_t* nze, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !src || !dst_row_p || !nblks || !nze || src->nblkrows != src->nblkcols || mode < 0 || mode > 2) return -1;
  hipStream_t st = stream_of(stream);
  const int nbr = src->nblkrows, W = (nbr + 31) / 32;
  E->valid = false;  // shares workspace with the symbolic phase
  *nblks = *nze = 0;
  if (nbr == 0) return 0;
  if (E->c_bm.ensure((size_t)nbr * W + 1) || E->c_pre.ensure((size_t)nbr * W + 1) || E->row_nnz.ensure((size_t)nbr + 1) ||
      E->blk_nze.ensure(2 * (size_t)src->nblks + 1) || E->c_blk_p_ws.ensure(2 * (size_t)src->nblks + 1) || E->dev_scalars.ensure(16))
    return -1;
  int64_t* dsc = reinterpret_cast<int64_t*>(E->dev_scalars.p);
  ACC_CHECK(hipMemsetAsync(E->c_bm.p, 0, sizeof(uint32_t) * (size_t)nbr * W, st));
  hipLaunchKernelGGL(desym_mark, grid_for((int64_t)nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, nbr, W, mode, E->c_bm.p);
  hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->c_bm.p, nbr, W, E->c_pre.p, E->row_nnz.p);
  if (exclusive_scan<int32_t>(E, E->row_nnz.p, nbr, dst_row_p, dsc + 0, true, st)) return -1;
  // block sizes in index order (square matrix: the transposed-matrix helper with rows = columns = the same sizes)
  hipLaunchKernelGGL(transpose_sizes, grid_for((int64_t)nbr * W), dim3(256), 0, st, E->c_bm.p, E->c_pre.p, dst_row_p, src->row_blk_size,
                     src->col_blk_size, nbr, W, E->blk_nze.p);
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  *nblks = E->host_scalars[0];
  if (exclusive_scan<int64_t>(E, E->blk_nze.p, *nblks, E->c_blk_p_ws.p, dsc + 1, false, st)) return -1;
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  *nze = E->host_scalars[1];
  return check(hipGetLastError(), "dbcsr_amd_bcsr_twin_count", __FILE__, __LINE__);
}

int dbcsr_amd_bcsr_twin_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, int mode, int antisymmetric, dbcsr_amd_bcsr* dst,
                              void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) plan_invalidate(E);  // this call uses (or changes what feeds) the engine's work areas: the next multiply runs its own symbolic phase
  if (!E || !src || !dst || src->nblkrows != src->nblkcols || mode < 0 || mode > 2) return -1;
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  hipStream_t st = stream_of(stream);
  const int nbr = src->nblkrows, W = (nbr + 31) / 32;
  if (nbr == 0 || src->nblks == 0) return 0;
  if (datatype == dbcsr_type_real_8)
    hipLaunchKernelGGL((desym_fill<double>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const double*>(src->data), src->row_blk_size, E->c_bm.p, E->c_pre.p, dst->row_p, E->c_blk_p_ws.p, nbr, W,
                       antisymmetric ? -1.0 : 1.0, mode, dst->col_i, dst->blk_p, static_cast<double*>(dst->data));
  else
    hipLaunchKernelGGL((desym_fill<float>), grid_for((int64_t)nbr * 64), dim3(256), 0, st, src->row_p, src->col_i, src->blk_p,
                       static_cast<const float*>(src->data), src->row_blk_size, E->c_bm.p, E->c_pre.p, dst->row_p, E->c_blk_p_ws.p, nbr, W,
                       antisymmetric ? -1.0f : 1.0f, mode, dst->col_i, dst->blk_p, static_cast<float*>(dst->data));
  return check(hipGetLastError(), "dbcsr_amd_bcsr_twin_apply", __FILE__, __LINE__);
}

int dbcsr_amd_mm_stats(void* handle, dbcsr_amd_mnk_stat* out, int max_entries, int* n_entries, void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !n_entries || (max_entries > 0 && !out)) return -1;
  *n_entries = 0;
  if (!E->valid || !E->timed || E->c_nblks == 0) return 0;  // no numeric call yet / nothing to count
  hipStream_t st = stream_of(stream);
  if (E->stat_table.ensure(2 * (size_t)kStatSlots + 2)) return -1;
  unsigned long long* keys = E->stat_table.p;
  unsigned long long* counts = keys + kStatSlots;
  int* overflow = reinterpret_cast<int*>(counts + kStatSlots);
  ACC_CHECK(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * (2 * (size_t)kStatSlots + 2), st));
  hipLaunchKernelGGL(mnk_histogram, grid_for(E->c_nblks), dim3(256), 0, st, E->descs.p, E->c_nblks, E->entries.p, keys, counts, overflow);
  std::vector<unsigned long long> host(2 * (size_t)kStatSlots + 2);
  ACC_CHECK(hipMemcpyAsync(host.data(), keys, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost, st));
  ACC_CHECK(hipStreamSynchronize(st));
  if (*reinterpret_cast<const int*>(&host[2 * (size_t)kStatSlots])) {
    fprintf(stderr, "dbcsr_amd_mm_stats: more than %d distinct (m, n, k) triples\n", kStatSlots);
    return -1;
  }
  std::vector<dbcsr_amd_mnk_stat> all;
  for (int i = 0; i < kStatSlots; ++i)
    if (host[i]) {
      dbcsr_amd_mnk_stat r;
      r.m = (int32_t)(host[i] & 0xffffu);
      r.n = (int32_t)((host[i] >> 16) & 0xffffu);
      r.k = (int32_t)((host[i] >> 32) & 0x7fffffffu);
      r.reserved = 0;
      r.nproducts = (int64_t)host[kStatSlots + i];
      r.flop = 2ll * r.m * r.n * r.k * r.nproducts;
      all.push_back(r);
    }
  std::sort(all.begin(), all.end(), [](const dbcsr_amd_mnk_stat& a, const dbcsr_amd_mnk_stat& b) {
    return a.flop != b.flop ? a.flop > b.flop : (a.m != b.m ? a.m < b.m : (a.n != b.n ? a.n < b.n : a.k < b.k));
  });
  *n_entries = (int)all.size();
  for (int i = 0; i < (int)all.size() && i < max_entries; ++i) out[i] = all[i];
  return 0;
}

const char* dbcsr_amd_mm_kernel_name(libsmm_acc_data_t datatype) {
  return datatype == dbcsr_type_real_4 ? "mm_numeric_f32" : "mm_numeric_f64";
}

const char* dbcsr_amd_mm_last_kernel(void* handle) {
  Engine* E = static_cast<Engine*>(handle);
  return E ? E->last_kernel : "";
}

int dbcsr_amd_mm_plan_stats(void* handle, int64_t* reused, int64_t* built) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E) return -1;
  if (reused) *reused = E->plan_hits;
  if (built) *built = E->plan_misses;
  return 0;
}

int dbcsr_amd_mm_tile_stats(void* handle, int* waves_gave_up, int* list_mismatches) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E) return -1;
  if (strncmp(E->last_kernel, "mm_numeric_f64_tile", 19) != 0 || !E->tile_flags.p) return 1;
  int h[4] = {0, 0, 0, 0};
  ACC_CHECK(hipDeviceSynchronize());
  ACC_CHECK(hipMemcpy(h, E->tile_flags.p, sizeof h, hipMemcpyDeviceToHost));
  if (waves_gave_up) *waves_gave_up = h[0];
  if (list_mismatches) *list_mismatches = h[1];
  if (getenv("DBCSR_AMD_MM_TILE_VERBOSE"))
    fprintf(stderr, "dbcsr_amd tile kernel: %d waves gave up, %lld reads of the team counters, %lld products waited for the window (of %lld)\n", h[0],
            16ll * h[2], 16ll * h[3], (long long)E->nproducts);
  return 0;
}

}  // extern "C"
