// mm_aux.h -- checksum, synthetic fill, transpose, desymmetrize, block filter, crop / window scale, (m, n, k) statistics
// Part of the device-resident multiply engine: included by mm_engine.hip (one translation unit), in this order:
// mm_workspace.h, mm_symbolic.h, mm_numeric_f64.h, mm_numeric_f32.h, mm_aux.h.
#ifndef DBCSR_AMD_MM_AUX_H
#define DBCSR_AMD_MM_AUX_H

namespace dbcsr_amd {

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

__global__ void store_scalar_f64(double* __restrict__ p, double v) { *p = v; }

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
#endif
