/*
 * dbcsr_oracle.c -- CPU restatement of the DBCSR block-sparse multiply hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the parity oracle (and the timed
 * "cpu_baseline" leg of bench.py).  Nothing in the product path (dbcsr_amd/)
 * may import, call or link it.
 *
 * Pinning: the restatement is checked against the reference's own golden
 * checksums (tests/inputs/[name].perf: checksum and position checksum with
 * threshold 1e-11) in tests/test_oracle_golden.py, and the LAPACK dlarnv
 * restatement against scipy's bundled OpenBLAS (scipy_dlarnv_) in
 * tests/test_oracle_larnv.py.
 *
 * Each function cites the reference file:line whose BEHAVIOUR it restates
 * (paths relative to /root/reference).  No reference source is copied.
 *
 * Conventions used here (all 0-based, 64-bit offsets; the reference is 1-based
 * int32): a BCSR matrix is (row_p[nbr+1], col_i[nblk], blk_p[nblk], data[]),
 * block (r,c) is row_size[r] x col_size[c], column-major, starting at
 * data[blk_p[.]].
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(_OPENMP)
#  include <omp.h>
#endif

typedef int64_t i64;
typedef uint64_t u64;

/* ------------------------------------------------------------------------- */
/* LAPACK xLARNV(idist=1) / xLARUV.                                           */
/* Third-party dependency of the reference (LAPACK, any vendor; not vendored  */
/* in /root/reference).  Published algorithm (LAPACK 3.x dlaruv.f): 48-bit    */
/* multiplicative congruential generator, modulus 2^48, multiplier            */
/* a = 33952834046453; the i-th number of a call is seed*a^i mod 2^48, the    */
/* seed after the call is the last number generated.  The 48-bit state is     */
/* held as four 12-bit limbs iseed[0..3] (iseed[0] most significant).         */
/* Reference call sites: src/utils/dbcsr_blas_operations.F:54-76,             */
/* src/ops/dbcsr_test_methods.F:397, 427.                                     */
/* ------------------------------------------------------------------------- */
#define LARUV_A ((u64)33952834046453ULL)
#define MASK48 ((((u64)1) << 48) - 1)

static u64 seed_pack(const int s[4]) {
  return (((u64)s[0]) << 36) | (((u64)s[1]) << 24) | (((u64)s[2]) << 12) | ((u64)s[3]);
}
static void seed_unpack(u64 x, int s[4]) {
  s[0] = (int)((x >> 36) & 4095);
  s[1] = (int)((x >> 24) & 4095);
  s[2] = (int)((x >> 12) & 4095);
  s[3] = (int)(x & 4095);
}

/* dlarnv(idist=1): uniform (0,1), double.  R*(it1+R*(it2+R*(it3+R*it4))) is
 * exact in double (48 < 53 bits), i.e. x * 2^-48; it can never round to 1. */
void orc_dlarnv1(int iseed[4], i64 n, double* x) {
  u64 s = seed_pack(iseed);
  for (i64 i = 0; i < n; ++i) {
    s = (s * LARUV_A) & MASK48;
    x[i] = (double)s * (1.0 / 281474976710656.0);
  }
  seed_unpack(s, iseed);
}

/* slarnv(idist=1): uniform (0,1), single.  slaruv evaluates the limb
 * polynomial in REAL arithmetic and, when it rounds to exactly 1.0, bumps
 * every limb of the call's base seed by 2 and retries (LAPACK >= 3.2
 * slaruv.f).  slarnv draws in chunks of 64 (LV/2). */
void orc_slarnv1(int iseed[4], i64 n, float* x) {
  const float r = 1.0f / 4096.0f;
  i64 done = 0;
  while (done < n) {
    const int il = (int)((n - done) < 64 ? (n - done) : 64);
    int i1 = iseed[0], i2 = iseed[1], i3 = iseed[2], i4 = iseed[3];
    int it1 = 0, it2 = 0, it3 = 0, it4 = 0;
    u64 apow = 1;
    for (int i = 0; i < il; ++i) {
      apow = (apow * LARUV_A) & MASK48;
      for (;;) {
        /* limb arithmetic of slaruv is a 48-bit truncated product; limbs that
         * overflowed 4095 after a +2 bump still enter as plain integers */
        const u64 full = ((((u64)i1) << 36) + (((u64)i2) << 24) + (((u64)i3) << 12) + (u64)i4);
        const u64 p = (full * apow) & MASK48;
        it1 = (int)((p >> 36) & 4095);
        it2 = (int)((p >> 24) & 4095);
        it3 = (int)((p >> 12) & 4095);
        it4 = (int)(p & 4095);
        const float v = r * ((float)it1 + r * ((float)it2 + r * ((float)it3 + r * (float)it4)));
        if (v == 1.0f) {
          i1 += 2;
          i2 += 2;
          i3 += 2;
          i4 += 2;
          continue;
        }
        x[done + i] = v;
        break;
      }
    }
    iseed[0] = it1;
    iseed[1] = it2;
    iseed[2] = it3;
    iseed[3] = it4;
    done += il;
  }
}

/* src/utils/dbcsr_blas_operations.F:29-52 (set_larnv_seed): a LAPACK-legal
 * seed that is a pure function of (irow, icol, nrow, ival). irow/icol 1-based. */
void orc_set_larnv_seed(int irow, int nrow, int icol, int ncol, int ival, int iseed[4]) {
  (void)ncol;
  i64 ivm = ((i64)ival) % 65536;
  if (ivm < 0) ivm += 65536;
  i64 map = (((i64)irow - 1 + (i64)icol * (i64)nrow) * (1 + ivm)) * 2 + 1;
  iseed[3] = (int)(map % 4096);
  map /= 4096;
  iseed[2] = (int)((map ^ 3541) % 4096);
  map /= 4096;
  iseed[1] = (int)((map ^ 1153) % 4096);
  map /= 4096;
  iseed[0] = (int)((map ^ 2029) % 4096);
}

/* ------------------------------------------------------------------------- */
/* Synthetic inputs: src/ops/dbcsr_test_methods.F                             */
/* ------------------------------------------------------------------------- */

/* :467-515 dbcsr_make_random_block_sizes: cycle through (multiplicity,size)
 * pairs, truncating the last block so the sizes sum to size_sum. Returns the
 * number of blocks (writes at most cap entries). */
int orc_make_block_sizes(int size_sum, const int* mix, int npairs, int* out, int cap) {
  int nblocks = 0, cur = 0, sel = 0, rep = 1;
  while (cur < size_sum) {
    int bs = mix[2 * sel + 1];
    if (bs > size_sum - cur) bs = size_sum - cur;
    if (nblocks < cap) out[nblocks] = bs;
    ++nblocks;
    cur += bs;
    ++rep;
    if (rep > mix[2 * sel]) {
      rep = 1;
      sel = (sel + 1) % npairs;
    }
  }
  return nblocks;
}

/* :388-411 block-presence pattern: geometric jumps driven by one dlarnv
 * stream seeded with seed(7,42,3,42,counter). rows/cols are 0-based and come
 * out sorted row-major.  Returns the number of blocks (writes <= cap). */
i64 orc_random_pattern(int nrow, int ncol, double sparsity, int counter, int* rows, int* cols, i64 cap) {
  int jseed[4];
  double my_sparsity = sparsity > 1.0 ? sparsity / 100.0 : sparsity;
  const i64 nmax = (i64)nrow * (i64)ncol;
  i64 ele = -1, cnt = 0;
  orc_set_larnv_seed(7, 42, 3, 42, counter, jseed);
  const double lsp = my_sparsity > 0 ? log(my_sparsity) : 0.0;
  for (;;) {
    double v;
    i64 inc;
    orc_dlarnv1(jseed, 1, &v);
    if (my_sparsity > 0)
      inc = 1 + (i64)floor(log(v) / lsp);
    else
      inc = 1;
    ele += inc;
    if (ele >= nmax) break;
    if (cnt < cap) {
      rows[cnt] = (int)(ele / ncol);
      cols[cnt] = (int)(ele % ncol);
    }
    ++cnt;
  }
  return cnt;
}

/* :423-429 block values: dlarnv(1, seed(row,nrow,col,ncol,counter), m*n),
 * column-major -- a pure function of (row, col, counter). */
void orc_fill_blocks_d(i64 nblks, const int* rows, const int* cols, int nrow, int ncol, int counter, const int* row_sizes,
  const int* col_sizes, const i64* blk_p, double* data) {
#pragma omp parallel for schedule(dynamic, 64)
  for (i64 b = 0; b < nblks; ++b) {
    int iseed[4];
    orc_set_larnv_seed(rows[b] + 1, nrow, cols[b] + 1, ncol, counter, iseed);
    orc_dlarnv1(iseed, (i64)row_sizes[rows[b]] * col_sizes[cols[b]], data + blk_p[b]);
  }
}
void orc_fill_blocks_s(i64 nblks, const int* rows, const int* cols, int nrow, int ncol, int counter, const int* row_sizes,
  const int* col_sizes, const i64* blk_p, float* data) {
#pragma omp parallel for schedule(dynamic, 64)
  for (i64 b = 0; b < nblks; ++b) {
    int iseed[4];
    orc_set_larnv_seed(rows[b] + 1, nrow, cols[b] + 1, ncol, counter, iseed);
    orc_slarnv1(iseed, (i64)row_sizes[rows[b]] * col_sizes[cols[b]], data + blk_p[b]);
  }
}

/* ------------------------------------------------------------------------- */
/* Checksums: src/dist/dbcsr_dist_util.F:432-577                              */
/*   checksum = sum x^2 ; pos = sum x * ln|grow * gcol| (1-based coordinates) */
/* Summation order as the reference: per block, then per block row, then over */
/* rows.                                                                      */
/* ------------------------------------------------------------------------- */
static void offsets_of(const int* sizes, int n, i64* off) {
  i64 o = 0;
  for (int i = 0; i < n; ++i) {
    off[i] = o;
    o += sizes[i];
  }
}

double orc_checksum_d(int nbr, int nbc, const int* row_sizes, const int* col_sizes, const int* row_p, const int* col_i,
  const i64* blk_p, const double* data, int pos) {
  i64* roff = (i64*)malloc(sizeof(i64) * (size_t)(nbr > 0 ? nbr : 1));
  i64* coff = (i64*)malloc(sizeof(i64) * (size_t)(nbc > 0 ? nbc : 1));
  offsets_of(row_sizes, nbr, roff);
  offsets_of(col_sizes, nbc, coff);
  double total = 0.0;
  for (int br = 0; br < nbr; ++br) {
    const int m = row_sizes[br];
    double rowsum = 0.0;
    for (int blk = row_p[br]; blk < row_p[br + 1]; ++blk) {
      const int bc = col_i[blk];
      const int n = col_sizes[bc];
      const double* d = data + blk_p[blk];
      double cs = 0.0;
      if (pos) {
        for (int c = 0; c < n; ++c)
          for (int r = 0; r < m; ++r)
            cs += d[(i64)c * m + r] * log(fabs((double)(roff[br] + r + 1) * (double)(coff[bc] + c + 1)));
      }
      else {
        for (i64 e = 0; e < (i64)m * n; ++e) cs += d[e] * d[e];
      }
      rowsum += cs;
    }
    total += rowsum;
  }
  free(roff);
  free(coff);
  return total;
}
double orc_checksum_s(int nbr, int nbc, const int* row_sizes, const int* col_sizes, const int* row_p, const int* col_i,
  const i64* blk_p, const float* data, int pos) {
  i64* roff = (i64*)malloc(sizeof(i64) * (size_t)(nbr > 0 ? nbr : 1));
  i64* coff = (i64*)malloc(sizeof(i64) * (size_t)(nbc > 0 ? nbc : 1));
  offsets_of(row_sizes, nbr, roff);
  offsets_of(col_sizes, nbc, coff);
  double total = 0.0;
  for (int br = 0; br < nbr; ++br) {
    const int m = row_sizes[br];
    double rowsum = 0.0;
    for (int blk = row_p[br]; blk < row_p[br + 1]; ++blk) {
      const int bc = col_i[blk];
      const int n = col_sizes[bc];
      const float* d = data + blk_p[blk];
      double cs = 0.0;
      if (pos) {
        for (int c = 0; c < n; ++c)
          for (int r = 0; r < m; ++r)
            cs += (double)d[(i64)c * m + r] * log(fabs((double)(roff[br] + r + 1) * (double)(coff[bc] + c + 1)));
      }
      else {
        /* DOT_PRODUCT of r_sp with itself is evaluated in single precision */
        float dot = 0.0f;
        for (i64 e = 0; e < (i64)m * n; ++e) dot += d[e] * d[e];
        cs = (double)dot;
      }
      rowsum += cs;
    }
    total += rowsum;
  }
  free(roff);
  free(coff);
  return total;
}

/* ------------------------------------------------------------------------- */
/* Stack level: the libsmm_acc ABI semantics                                  */
/* ------------------------------------------------------------------------- */

/* CPU executor of one homogeneous parameter stack.
 * b_transposed=0: src/mm/dbcsr_mm_hostdrv.F:248-282 (blas_process_mm_stack:
 *   DGEMM('N','N',m,n,k,1,A(a),m,B(b),k,1,C(c),m), entries in stack order).
 * b_transposed=1: the layout the GPU kernels see after libsmm_acc_transpose,
 *   src/acc/libsmm_acc/libsmm_acc_benchmark.cpp:126-145 (B stored n x k).
 * stack = 3 ints per entry: 1-based element offsets (a, b, c). */
void orc_stack_calc_d(
  const int* stack, int nstack, double* c, const double* a, const double* b, int m, int n, int k, int b_transposed) {
  for (int s = 0; s < nstack; ++s) {
    const double* A = a + (stack[3 * s] - 1);
    const double* B = b + (stack[3 * s + 1] - 1);
    double* C = c + (stack[3 * s + 2] - 1);
    for (int nn = 0; nn < n; ++nn)
      for (int mm = 0; mm < m; ++mm) {
        double res = 0.0;
        for (int kk = 0; kk < k; ++kk) res += A[(i64)kk * m + mm] * (b_transposed ? B[(i64)kk * n + nn] : B[(i64)nn * k + kk]);
        C[(i64)nn * m + mm] += res;
      }
  }
}
void orc_stack_calc_s(
  const int* stack, int nstack, float* c, const float* a, const float* b, int m, int n, int k, int b_transposed) {
  for (int s = 0; s < nstack; ++s) {
    const float* A = a + (stack[3 * s] - 1);
    const float* B = b + (stack[3 * s + 1] - 1);
    float* C = c + (stack[3 * s + 2] - 1);
    for (int nn = 0; nn < n; ++nn)
      for (int mm = 0; mm < m; ++mm) {
        float res = 0.0f;
        for (int kk = 0; kk < k; ++kk) res += A[(i64)kk * m + mm] * (b_transposed ? B[(i64)kk * n + nn] : B[(i64)nn * k + kk]);
        C[(i64)nn * m + mm] += res;
      }
  }
}

/* Inhomogeneous stack, 7 ints per entry (m,n,k,a,b,c,c_blk), 1-based offsets:
 * src/mm/dbcsr_mm_types.F:24-37 + dbcsr_mm_hostdrv.F:248-282. */
void orc_stack7_calc_d(const int* stack7, int nstack, double* c, const double* a, const double* b) {
  for (int s = 0; s < nstack; ++s) {
    const int* p = stack7 + 7 * s;
    const int m = p[0], n = p[1], k = p[2];
    const double* A = a + (p[3] - 1);
    const double* B = b + (p[4] - 1);
    double* C = c + (p[5] - 1);
    for (int nn = 0; nn < n; ++nn)
      for (int kk = 0; kk < k; ++kk) {
        const double bv = B[(i64)nn * k + kk];
        for (int mm = 0; mm < m; ++mm) C[(i64)nn * m + mm] += A[(i64)kk * m + mm] * bv;
      }
  }
}

/* In-place block transposes: src/acc/libsmm_acc/kernels/smm_acc_transpose.h:41-64
 * (block stored m x n column-major becomes n x m column-major); trs_stack holds
 * 0-based element offsets (src/mm/dbcsr_mm_common.F:412,429). */
void orc_transpose_d(const int* trs_stack, int nblks, double* data, int m, int n) {
  double* buf = (double*)malloc(sizeof(double) * (size_t)(m * n > 0 ? m * n : 1));
  for (int s = 0; s < nblks; ++s) {
    double* blk = data + trs_stack[s];
    memcpy(buf, blk, sizeof(double) * (size_t)(m * n));
    for (int i = 0; i < m * n; ++i) {
      const int r_out = i % n, c_out = i / n;
      blk[i] = buf[r_out * m + c_out];
    }
  }
  free(buf);
}

/* Squared Frobenius norm per block, fp64 in, fp32 out:
 * src/acc/cuda_hip/calculate_norms.cpp:47-118. */
void orc_norms_d(const double* mat, int nblks, const int* offsets, const int* nelems, float* norms) {
  for (int b = 0; b < nblks; ++b) {
    double s = 0.0;
    for (int i = 0; i < nelems[b]; ++i) s += mat[offsets[b] + i] * mat[offsets[b] + i];
    norms[b] = (float)s;
  }
}

/* Benchmark inputs of the reference's kernel validator / tuner:
 * src/acc/libsmm_acc/libsmm_acc_benchmark.cpp:103-109 (matInit) and
 * src/acc/acc_bench.h:48-79 (INIT_STACK with rnd==NULL -> libc rand()). */
void orc_mat_init(double* mat, int mat_n, int x, int y, int seed) {
  double* p = mat;
  for (int n = 0; n < mat_n; ++n)
    for (int j = 0; j < y; ++j)
      for (int i = 0; i < x; ++i, ++p) *p = (double)j * x + i + n + seed;
}
void orc_stack_init(int* stack, int nstack, int nc, int na, int nb, int m, int n, int k, unsigned rseed, int reseed) {
  const int mn = m * n, mk = m * k, kn = k * n;
  const int navg = nstack / nc;
  const int nimb = (navg - 4) > 1 ? (navg - 4) : 1;
  int i = 0, c = 0, ntop = 0;
  int* p = stack;
  if (reseed) srand(rseed);
  while (i < nstack) {
    const int r = rand();
    const int next = c + 1;
    ntop += navg + (r % (2 * nimb) - nimb);
    if (nstack < ntop) ntop = nstack;
    for (; i < ntop; ++i) {
      const int ia = rand() % na;
      const int ib = rand() % nb;
      *p++ = ia * mk + 1;
      *p++ = ib * kn + 1;
      *p++ = c * mn + 1;
    }
    if (next < nc) c = next;
  }
}
double orc_check_sum(const double* mat, i64 n) {
  double r = 0;
  for (i64 i = 0; i < n; ++i) r += mat[i];
  return r;
}

/* ------------------------------------------------------------------------- */
/* The multiply: C <- beta*C + alpha*op(A)*op(B)                              */
/*                                                                            */
/*  src/mm/dbcsr_mm.F:336-1023 (dbcsr_multiply_generic: op(), beta scaling,   */
/*    alpha folded into one operand, work matrix seeded with C's blocks),     */
/*  src/mm/dbcsr_mm_csr.F:178-359 (symbolic product, new C block gets         */
/*    offset = datasize+1 in discovery order, filter on norms),               */
/*  src/mm/dbcsr_mm_hostdrv.F:248-282 (one GEMM per product),                 */
/*  src/mm/dbcsr_mm_multrec.F:694-748 (final block filter),                   */
/*  src/work/dbcsr_work_operations.F:749+ (final index sorted by row, col).   */
/*                                                                            */
/* The recursion of dbcsr_mm_multrec.F:487-576 only re-orders the products    */
/* (cache blocking); it is not restated -- products of a C block are summed   */
/* in ascending k, which differs from the reference in rounding only.         */
/* ------------------------------------------------------------------------- */

typedef struct {
  int nbr, nbc;
  i64 nblks, nze;
  int* row_p;
  int* col_i;
  i64* blk_p; /* sorted (row, col) order, data compacted in that order */
  double* data;
  /* discovery-order view, as the reference's work matrix holds it */
  int* disc_row;
  int* disc_col;
  i64 flop;
  i64 nproducts;
} orc_result;

static void transpose_bcsr_d(int nbr, int nbc, const int* rs, const int* cs, const int* row_p, const int* col_i, const i64* blk_p,
  const double* data, int** t_row_p, int** t_col_i, i64** t_blk_p, double** t_data) {
  /* dbcsr_new_transposed (src/ops/dbcsr_transformations.F): block (r,c) m x n
   * becomes block (c,r) n x m with transposed data */
  const i64 nblks = row_p[nbr];
  int* trp = (int*)calloc((size_t)nbc + 1, sizeof(int));
  int* tci = (int*)malloc(sizeof(int) * (size_t)(nblks > 0 ? nblks : 1));
  i64* tbp = (i64*)malloc(sizeof(i64) * (size_t)(nblks > 0 ? nblks : 1));
  i64 nze = 0;
  for (int r = 0; r < nbr; ++r)
    for (int b = row_p[r]; b < row_p[r + 1]; ++b) {
      trp[col_i[b] + 1]++;
      nze += (i64)rs[r] * cs[col_i[b]];
    }
  for (int c = 0; c < nbc; ++c) trp[c + 1] += trp[c];
  double* td = (double*)malloc(sizeof(double) * (size_t)(nze > 0 ? nze : 1));
  int* fill = (int*)calloc((size_t)nbc + 1, sizeof(int));
  /* first assign slots (sorted by new row = old col, then new col = old row) */
  for (int r = 0; r < nbr; ++r)
    for (int b = row_p[r]; b < row_p[r + 1]; ++b) {
      const int c = col_i[b];
      const int slot = trp[c] + fill[c]++;
      tci[slot] = r;
      tbp[slot] = b; /* temporarily remember source block */
    }
  i64 off = 0;
  for (i64 s = 0; s < nblks; ++s) {
    const int b = (int)tbp[s];
    /* find source row: tci[s] */
    const int r = tci[s];
    const int c = col_i[b];
    const int m = rs[r], n = cs[c];
    const double* src = data + blk_p[b];
    double* dst = td + off;
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < m; ++i) dst[(i64)i * n + j] = src[(i64)j * m + i];
    tbp[s] = off;
    off += (i64)m * n;
  }
  free(fill);
  *t_row_p = trp;
  *t_col_i = tci;
  *t_blk_p = tbp;
  *t_data = td;
}

static inline void block_gemm_acc(int m, int n, int k, double alpha, const double* A, const double* B, double* C) {
  for (int nn = 0; nn < n; ++nn)
    for (int kk = 0; kk < k; ++kk) {
      const double bv = alpha * B[(i64)nn * k + kk];
      const double* Ac = A + (i64)kk * m;
      double* Cc = C + (i64)nn * m;
      for (int mm = 0; mm < m; ++mm) Cc[mm] += Ac[mm] * bv;
    }
}

/* limits: first_row,last_row,first_col,last_col,first_k,last_k as 0-based BLOCK
 * indices, inclusive; -1 = no limit.  (The reference takes element limits and
 * requires them on block boundaries for the paths used here.) */
orc_result* orc_multiply_d(char transa, char transb, double alpha,
  /* A */ int a_nbr, int a_nbc, const int* a_rs, const int* a_cs, const int* a_row_p, const int* a_col_i, const i64* a_blk_p,
  const double* a_data,
  /* B */ int b_nbr, int b_nbc, const int* b_rs, const int* b_cs, const int* b_row_p, const int* b_col_i, const i64* b_blk_p,
  const double* b_data, double beta,
  /* C */ int c_nbr, int c_nbc, const int* c_rs, const int* c_cs, const int* c_row_p, const int* c_col_i, const i64* c_blk_p,
  const double* c_data, int retain_sparsity, double filter_eps, const int* limits, int canonical_c) {
  /* canonical_c: the product matrix has symmetry and its index is in canonical (checkerboard) form, dbcsr_mm.F:711-719: the
   * local multiply then leaves out the blocks whose twin (j, i) is the stored one (dbcsr_mm_csr.F:280-292) */
  int *ta_row_p = NULL, *ta_col_i = NULL, *tb_row_p = NULL, *tb_col_i = NULL;
  i64 *ta_blk_p = NULL, *tb_blk_p = NULL;
  double *ta_data = NULL, *tb_data = NULL;
  if (transa == 'T' || transa == 't' || transa == 'C' || transa == 'c') {
    transpose_bcsr_d(a_nbr, a_nbc, a_rs, a_cs, a_row_p, a_col_i, a_blk_p, a_data, &ta_row_p, &ta_col_i, &ta_blk_p, &ta_data);
    int t = a_nbr;
    a_nbr = a_nbc;
    a_nbc = t;
    const int* ts = a_rs;
    a_rs = a_cs;
    a_cs = ts;
    a_row_p = ta_row_p;
    a_col_i = ta_col_i;
    a_blk_p = ta_blk_p;
    a_data = ta_data;
  }
  if (transb == 'T' || transb == 't' || transb == 'C' || transb == 'c') {
    transpose_bcsr_d(b_nbr, b_nbc, b_rs, b_cs, b_row_p, b_col_i, b_blk_p, b_data, &tb_row_p, &tb_col_i, &tb_blk_p, &tb_data);
    int t = b_nbr;
    b_nbr = b_nbc;
    b_nbc = t;
    const int* ts = b_rs;
    b_rs = b_cs;
    b_cs = ts;
    b_row_p = tb_row_p;
    b_col_i = tb_col_i;
    b_blk_p = tb_blk_p;
    b_data = tb_data;
  }
  if (a_nbr != c_nbr || b_nbc != c_nbc || a_nbc != b_nbr) {
    fprintf(stderr, "orc_multiply_d: incompatible block dimensions\n");
    return NULL;
  }
  int lim[6] = {-1, -1, -1, -1, -1, -1};
  if (limits)
    for (int i = 0; i < 6; ++i) lim[i] = limits[i];
  const int r0 = lim[0] < 0 ? 0 : lim[0], r1 = lim[1] < 0 ? c_nbr - 1 : lim[1];
  const int j0 = lim[2] < 0 ? 0 : lim[2], j1 = lim[3] < 0 ? c_nbc - 1 : lim[3];
  const int k0 = lim[4] < 0 ? 0 : lim[4], k1 = lim[5] < 0 ? a_nbc - 1 : lim[5];

  /* block norms for on-the-fly filtering (dbcsr_mm_csr.F:276, fp32 norms of
   * the squared Frobenius norm; dbcsr_mm_cannon.F:1040-1113 row eps) */
  const int use_eps = filter_eps > 0.0;
  float *a_norms = NULL, *b_norms = NULL, *row_eps = NULL;
  if (use_eps) {
    const i64 na = a_row_p[a_nbr], nb = b_row_p[b_nbr];
    a_norms = (float*)malloc(sizeof(float) * (size_t)(na > 0 ? na : 1));
    b_norms = (float*)malloc(sizeof(float) * (size_t)(nb > 0 ? nb : 1));
    row_eps = (float*)malloc(sizeof(float) * (size_t)(a_nbr > 0 ? a_nbr : 1));
    for (int r = 0; r < a_nbr; ++r) {
      for (int b = a_row_p[r]; b < a_row_p[r + 1]; ++b) {
        double s = 0;
        const i64 ne = (i64)a_rs[r] * a_cs[a_col_i[b]];
        for (i64 e = 0; e < ne; ++e) s += a_data[a_blk_p[b] + e] * a_data[a_blk_p[b] + e];
        a_norms[b] = (float)s;
      }
      /* dbcsr_mm_cannon.F:1100-1110: row_max_epss = (eps / max(1,row blocks))^2 */
      const int cnt = a_row_p[r + 1] - a_row_p[r];
      const float e = (float)filter_eps / (float)(cnt > 1 ? cnt : 1); /* single precision as the reference */
      row_eps[r] = e * e;
    }
    for (int r = 0; r < b_nbr; ++r)
      for (int b = b_row_p[r]; b < b_row_p[r + 1]; ++b) {
        double s = 0;
        const i64 ne = (i64)b_rs[r] * b_cs[b_col_i[b]];
        for (i64 e = 0; e < ne; ++e) s += alpha * b_data[b_blk_p[b] + e] * alpha * b_data[b_blk_p[b] + e];
        b_norms[b] = (float)s;
      }
  }

  /* keep_product_data (dbcsr_mm.F:695-704): the old C is kept when retain_sparsity, beta != 0, or a row / column window
   * ends inside C; otherwise the reference empties the product matrix before multiplying (dbcsr_mm.F:865-870) -- its
   * blocks disappear and their values are never read. */
  const int keep_product_data = retain_sparsity || beta != 0.0 || (lim[1] >= 0 && lim[1] < c_nbr - 1) || (lim[3] >= 0 && lim[3] < c_nbc - 1);
  int* zero_row_p = NULL;
  if (!keep_product_data) {
    zero_row_p = (int*)calloc((size_t)c_nbr + 1, sizeof(int));
    c_row_p = zero_row_p;
  }

  orc_result* R = (orc_result*)calloc(1, sizeof(orc_result));
  R->nbr = c_nbr;
  R->nbc = c_nbc;

  /* Work matrix, discovery order.  Existing C blocks come first (beta-scaled),
   * dbcsr_mm.F:706-709, 821-844. */
  i64 cap_blk = (c_row_p[c_nbr] > 0 ? c_row_p[c_nbr] : 16) * 2;
  i64 cap_dat = 1024;
  {
    i64 nze = 0;
    for (int r = 0; r < c_nbr; ++r)
      for (int b = c_row_p[r]; b < c_row_p[r + 1]; ++b) nze += (i64)c_rs[r] * c_cs[c_col_i[b]];
    cap_dat = nze * 2 + 1024;
  }
  int* w_row = (int*)malloc(sizeof(int) * (size_t)cap_blk);
  int* w_col = (int*)malloc(sizeof(int) * (size_t)cap_blk);
  i64* w_off = (i64*)malloc(sizeof(i64) * (size_t)cap_blk);
  double* w_dat = (double*)malloc(sizeof(double) * (size_t)cap_dat);
  i64 lastblk = 0, datasize = 0;
  /* per-row lookup of C blocks (the reference uses one hash table per row,
   * src/utils/dbcsr_hash_table.f90; a dense map has the same semantics) */
  i64* lut = (i64*)malloc(sizeof(i64) * (size_t)(c_nbc > 0 ? c_nbc : 1));
  for (int j = 0; j < c_nbc; ++j) lut[j] = -1;

  /* row pointers into the work list for the pre-existing blocks */
  i64* pre_start = (i64*)malloc(sizeof(i64) * ((size_t)c_nbr + 1));
  for (int r = 0; r < c_nbr; ++r) {
    pre_start[r] = lastblk;
    for (int b = c_row_p[r]; b < c_row_p[r + 1]; ++b) {
      const i64 ne = (i64)c_rs[r] * c_cs[c_col_i[b]];
      w_row[lastblk] = r;
      w_col[lastblk] = c_col_i[b];
      w_off[lastblk] = datasize;
      for (i64 e = 0; e < ne; ++e) w_dat[datasize + e] = beta * c_data[c_blk_p[b] + e];
      datasize += ne;
      ++lastblk;
    }
  }
  pre_start[c_nbr] = lastblk;
  const i64 npre = lastblk;

  i64 flop = 0, nprod = 0;
  for (int i = r0; i <= r1; ++i) {
    const int m = a_rs[i];
    /* seed the row lookup with pre-existing blocks of this row */
    for (i64 w = pre_start[i]; w < pre_start[i + 1]; ++w) lut[w_col[w]] = w;
    const i64 row_first_new = lastblk;
    for (int ab = a_row_p[i]; ab < a_row_p[i + 1]; ++ab) {
      const int kb = a_col_i[ab];
      if (kb < k0 || kb > k1) continue;
      const int k = a_cs[kb];
      for (int bb = b_row_p[kb]; bb < b_row_p[kb + 1]; ++bb) {
        const int j = b_col_i[bb];
        if (j < j0 || j > j1) continue;
        if (use_eps && a_norms[ab] * b_norms[bb] < row_eps[i]) continue;
        /* checker_tr (dbcsr_dist_operations.F:65-75) on the 1-based logical coordinates */
        if (canonical_c && i != j && ((((i + j) & 1) == 1) == (j >= i))) continue;
        const int n = b_cs[j];
        i64 w = lut[j];
        if (w < 0) {
          if (retain_sparsity) continue;
          const i64 ne = (i64)m * n;
          if (lastblk + 1 > cap_blk) {
            cap_blk *= 2;
            w_row = (int*)realloc(w_row, sizeof(int) * (size_t)cap_blk);
            w_col = (int*)realloc(w_col, sizeof(int) * (size_t)cap_blk);
            w_off = (i64*)realloc(w_off, sizeof(i64) * (size_t)cap_blk);
          }
          if (datasize + ne > cap_dat) {
            while (datasize + ne > cap_dat) cap_dat *= 2;
            w_dat = (double*)realloc(w_dat, sizeof(double) * (size_t)cap_dat);
          }
          w = lastblk++;
          w_row[w] = i;
          w_col[w] = j;
          w_off[w] = datasize;
          memset(w_dat + datasize, 0, sizeof(double) * (size_t)ne);
          datasize += ne;
          lut[j] = w;
        }
        block_gemm_acc(m, n, k, alpha, a_data + a_blk_p[ab], b_data + b_blk_p[bb], w_dat + w_off[w]);
        flop += 2 * (i64)m * n * k;
        ++nprod;
      }
    }
    for (i64 w = pre_start[i]; w < pre_start[i + 1]; ++w) lut[w_col[w]] = -1;
    for (i64 w = row_first_new; w < lastblk; ++w) lut[w_col[w]] = -1;
  }
  (void)npre;

  /* final filter (dbcsr_mm_multrec.F:694-748: drop blocks with ||blk||^2 < eps^2)
   * then sorted index (dbcsr_work_operations.F:749+). */
  char* keep = (char*)malloc((size_t)(lastblk > 0 ? lastblk : 1));
  i64 nkeep = 0;
  for (i64 w = 0; w < lastblk; ++w) {
    keep[w] = 1;
    if (use_eps && !retain_sparsity) { /* dbcsr_mm_multrec.F:373-383 */
      const i64 ne = (i64)c_rs[w_row[w]] * c_cs[w_col[w]];
      double s = 0;
      for (i64 e = 0; e < ne; ++e) s += w_dat[w_off[w] + e] * w_dat[w_off[w] + e];
      if (s < filter_eps * filter_eps) keep[w] = 0;
    }
    nkeep += keep[w];
  }
  /* counting sort by row; within a row sort by column (rows are short) */
  R->nblks = nkeep;
  R->row_p = (int*)calloc((size_t)c_nbr + 1, sizeof(int));
  R->col_i = (int*)malloc(sizeof(int) * (size_t)(nkeep > 0 ? nkeep : 1));
  R->blk_p = (i64*)malloc(sizeof(i64) * (size_t)(nkeep > 0 ? nkeep : 1));
  R->disc_row = (int*)malloc(sizeof(int) * (size_t)(nkeep > 0 ? nkeep : 1));
  R->disc_col = (int*)malloc(sizeof(int) * (size_t)(nkeep > 0 ? nkeep : 1));
  {
    i64 d = 0;
    for (i64 w = 0; w < lastblk; ++w)
      if (keep[w]) {
        R->row_p[w_row[w] + 1]++;
        R->disc_row[d] = w_row[w];
        R->disc_col[d] = w_col[w];
        ++d;
      }
  }
  for (int r = 0; r < c_nbr; ++r) R->row_p[r + 1] += R->row_p[r];
  i64* slot_src = (i64*)malloc(sizeof(i64) * (size_t)(nkeep > 0 ? nkeep : 1));
  int* fill = (int*)calloc((size_t)c_nbr + 1, sizeof(int));
  for (i64 w = 0; w < lastblk; ++w)
    if (keep[w]) {
      const int r = w_row[w];
      const int s = R->row_p[r] + fill[r]++;
      R->col_i[s] = w_col[w];
      slot_src[s] = w;
    }
  for (int r = 0; r < c_nbr; ++r) {
    /* insertion sort by column inside the row */
    for (int s = R->row_p[r] + 1; s < R->row_p[r + 1]; ++s) {
      const int cj = R->col_i[s];
      const i64 sj = slot_src[s];
      int t = s - 1;
      while (t >= R->row_p[r] && R->col_i[t] > cj) {
        R->col_i[t + 1] = R->col_i[t];
        slot_src[t + 1] = slot_src[t];
        --t;
      }
      R->col_i[t + 1] = cj;
      slot_src[t + 1] = sj;
    }
  }
  i64 nze = 0;
  for (int r = 0; r < c_nbr; ++r)
    for (int s = R->row_p[r]; s < R->row_p[r + 1]; ++s) nze += (i64)c_rs[r] * c_cs[R->col_i[s]];
  R->nze = nze;
  R->data = (double*)malloc(sizeof(double) * (size_t)(nze > 0 ? nze : 1));
  {
    i64 off = 0;
    for (int r = 0; r < c_nbr; ++r)
      for (int s = R->row_p[r]; s < R->row_p[r + 1]; ++s) {
        const i64 ne = (i64)c_rs[r] * c_cs[R->col_i[s]];
        memcpy(R->data + off, w_dat + w_off[slot_src[s]], sizeof(double) * (size_t)ne);
        R->blk_p[s] = off;
        off += ne;
      }
  }
  R->flop = flop;
  R->nproducts = nprod;

  free(fill);
  free(slot_src);
  free(keep);
  free(pre_start);
  free(lut);
  free(zero_row_p);
  free(w_row);
  free(w_col);
  free(w_off);
  free(w_dat);
  free(a_norms);
  free(b_norms);
  free(row_eps);
  free(ta_row_p);
  free(ta_col_i);
  free(ta_blk_p);
  free(ta_data);
  free(tb_row_p);
  free(tb_col_i);
  free(tb_blk_p);
  free(tb_data);
  return R;
}

i64 orc_result_nblks(const orc_result* r) { return r->nblks; }
i64 orc_result_nze(const orc_result* r) { return r->nze; }
i64 orc_result_flop(const orc_result* r) { return r->flop; }
i64 orc_result_nproducts(const orc_result* r) { return r->nproducts; }
void orc_result_copy(const orc_result* r, int* row_p, int* col_i, i64* blk_p, double* data, int* disc_row, int* disc_col) {
  if (row_p) memcpy(row_p, r->row_p, sizeof(int) * ((size_t)r->nbr + 1));
  if (col_i) memcpy(col_i, r->col_i, sizeof(int) * (size_t)r->nblks);
  if (blk_p) memcpy(blk_p, r->blk_p, sizeof(i64) * (size_t)r->nblks);
  if (data) memcpy(data, r->data, sizeof(double) * (size_t)r->nze);
  if (disc_row) memcpy(disc_row, r->disc_row, sizeof(int) * (size_t)r->nblks);
  if (disc_col) memcpy(disc_col, r->disc_col, sizeof(int) * (size_t)r->nblks);
}
void orc_result_free(orc_result* r) {
  if (!r) return;
  free(r->row_p);
  free(r->col_i);
  free(r->blk_p);
  free(r->data);
  free(r->disc_row);
  free(r->disc_col);
  free(r);
}

/* ------------------------------------------------------------------------- */
/* Timed CPU baseline ("port"): same enumeration as orc_multiply_d, OpenMP    */
/* over A block rows as the reference's thread distribution does              */
/* (src/mm/dbcsr_mm_cannon.F:1651-1704), rows [row_begin,row_end) only so     */
/* bench.py can time a bounded sample.  C structure is precomputed (sorted    */
/* BCSR, zero-initialised or holding beta*C_in); returns the flop count.      */
/* ------------------------------------------------------------------------- */
i64 orc_multiply_rows_d(int row_begin, int row_end, const int* a_rs, const int* a_cs, const int* a_row_p, const int* a_col_i,
  const i64* a_blk_p, const double* a_data, const int* b_cs, const int* b_row_p, const int* b_col_i, const i64* b_blk_p,
  const double* b_data, int c_nbc, const int* c_row_p, const int* c_col_i, const i64* c_blk_p, double* c_data, int nthreads) {
  i64 flop = 0;
#if defined(_OPENMP)
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel reduction(+ : flop)
  {
    i64* lut = (i64*)malloc(sizeof(i64) * (size_t)(c_nbc > 0 ? c_nbc : 1));
    for (int j = 0; j < c_nbc; ++j) lut[j] = -1;
#pragma omp for schedule(dynamic, 1)
    for (int i = row_begin; i < row_end; ++i) {
      const int m = a_rs[i];
      for (int s = c_row_p[i]; s < c_row_p[i + 1]; ++s) lut[c_col_i[s]] = c_blk_p[s];
      for (int ab = a_row_p[i]; ab < a_row_p[i + 1]; ++ab) {
        const int kb = a_col_i[ab];
        const int k = a_cs[kb];
        for (int bb = b_row_p[kb]; bb < b_row_p[kb + 1]; ++bb) {
          const int j = b_col_i[bb];
          const i64 co = lut[j];
          if (co < 0) continue;
          const int n = b_cs[j];
          block_gemm_acc(m, n, k, 1.0, a_data + a_blk_p[ab], b_data + b_blk_p[bb], c_data + co);
          flop += 2 * (i64)m * n * k;
        }
      }
      for (int s = c_row_p[i]; s < c_row_p[i + 1]; ++s) lut[c_col_i[s]] = -1;
    }
    free(lut);
  }
  return flop;
}

int orc_max_threads(void) {
#if defined(_OPENMP)
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* The default of one OpenMP thread per visible CPU oversubscribes a host whose cgroup grants fewer CPUs than it shows (a GPU
 * box: 256 visible, quota 16): the front-end (oracle.py) sets the team size to what the quota allows. */
void orc_set_threads(int n) {
#if defined(_OPENMP)
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
