// mm_api.hip -- dbcsr_multiply for one rank as ONE native call (include/dbcsr_amd_mm.h: dbcsr_amd_multiply).
//
// The orchestration of the reference's dbcsr_multiply_generic (src/mm/dbcsr_mm.F:336-1023) that matters on the hot path --
// op(A)/op(B) (:520-580), submatrix limits (:631-709: crop of the left matrix to (rows, k) and of the right one to
// (k, columns) in make_m2s, dbcsr_mm_cannon.F:194-214; beta acts on the window of C only), retain_sparsity, filter_eps
// (on-the-fly product filter + final block filter, dbcsr_mm_multrec.F:373-383) -- written on top of the primitives of
// dbcsr_amd_mm.h, so that a Fortran / C host needs one binding instead of re-implementing the sequence.  The Python
// mirror dbcsr_amd/multiply.py does the same with torch tensors as storage; here storage comes from the library's caching device allocator and the
// result is handed to the caller (dbcsr_amd_bcsr_release frees it).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../include/dbcsr_amd_mm.h"
#include "common.h"

using namespace dbcsr_amd;

namespace {

size_t elem_size(libsmm_acc_data_t dt) { return dt == dbcsr_type_real_8 ? 8 : 4; }

// a BCSR matrix whose four arrays were allocated here
struct Owned {
  dbcsr_amd_bcsr m;
  bool live = false;
  Owned() {
    m.nblkrows = m.nblkcols = 0;
    m.row_blk_size = m.col_blk_size = nullptr;
    m.row_p = m.col_i = nullptr;
    m.blk_p = nullptr;
    m.data = nullptr;
    m.nblks = 0;
    m.index_stamp = 0;
  }
  void release() {
    if (!live) return;
    (void)pool_free(m.row_p);
    (void)pool_free(m.col_i);
    (void)pool_free(m.blk_p);
    (void)pool_free(m.data);
    m.row_p = m.col_i = nullptr;
    m.blk_p = nullptr;
    m.data = nullptr;
    live = false;
  }
  ~Owned() { release(); }
  Owned(const Owned&) = delete;
  Owned& operator=(const Owned&) = delete;
};

int alloc_arrays(Owned& o, int nbr, int nbc, const int32_t* rs, const int32_t* cs, int64_t nblks, int64_t nze, size_t esz, bool with_row_p) {
  o.m.nblkrows = nbr;
  o.m.nblkcols = nbc;
  o.m.row_blk_size = rs;
  o.m.col_blk_size = cs;
  o.m.nblks = nblks;
  o.live = true;
  if (with_row_p && pool_malloc(reinterpret_cast<void**>(&o.m.row_p), sizeof(int32_t) * ((size_t)nbr + 1)) != hipSuccess) return -1;
  if (pool_malloc(reinterpret_cast<void**>(&o.m.col_i), sizeof(int32_t) * (size_t)(nblks > 0 ? nblks : 1)) != hipSuccess) return -1;
  if (pool_malloc(reinterpret_cast<void**>(&o.m.blk_p), sizeof(int64_t) * (size_t)(nblks > 0 ? nblks : 1)) != hipSuccess) return -1;
  if (pool_malloc(&o.m.data, esz * (size_t)(nze > 0 ? nze : 1)) != hipSuccess) return -1;
  return 0;
}

int alloc_row_p(Owned& o, int nbr) {
  o.live = true;
  return pool_malloc(reinterpret_cast<void**>(&o.m.row_p), sizeof(int32_t) * ((size_t)nbr + 1)) == hipSuccess ? 0 : -1;
}

// copy of src restricted to a window (negative bound = unbounded); also the way to learn nblks / nze of a matrix
int crop(void* h, libsmm_acc_data_t dt, const dbcsr_amd_bcsr* src, int64_t r0, int64_t r1, int64_t c0, int64_t c1, Owned& dst, int64_t* nze_out,
         void* stream) {
  if (alloc_row_p(dst, src->nblkrows)) return -1;
  int64_t nb = 0, nz = 0;
  int rc = dbcsr_amd_bcsr_crop_count(h, dt, src, r0, r1, c0, c1, dst.m.row_p, &nb, &nz, stream);
  if (rc) return rc;
  if (alloc_arrays(dst, src->nblkrows, src->nblkcols, src->row_blk_size, src->col_blk_size, nb, nz, elem_size(dt), false)) return -1;
  if (nze_out) *nze_out = nz;
  return dbcsr_amd_bcsr_crop_apply(h, dt, src, &dst.m, stream);
}

int transposed(void* h, libsmm_acc_data_t dt, const dbcsr_amd_bcsr* src, Owned& dst, void* stream) {
  // sizes of src: one counting pass over the whole matrix
  Owned probe;
  if (alloc_row_p(probe, src->nblkrows)) return -1;
  int64_t nb = 0, nz = 0;
  int rc = dbcsr_amd_bcsr_crop_count(h, dt, src, -1, -1, -1, -1, probe.m.row_p, &nb, &nz, stream);
  if (rc) return rc;
  if (alloc_arrays(dst, src->nblkcols, src->nblkrows, src->col_blk_size, src->row_blk_size, nb, nz, elem_size(dt), true)) return -1;
  return dbcsr_amd_bcsr_transpose(h, dt, src, &dst.m, stream);
}


// block (r, c) <-> its twin (c, r): mode 1 stored triangle -> canonical form, mode 2 canonical form -> stored triangle (dbcsr_amd_bcsr_twin_*)
int twin(void* h, libsmm_acc_data_t dt, const dbcsr_amd_bcsr* src, int mode, int antisymmetric, Owned& dst, void* stream) {
  if (alloc_row_p(dst, src->nblkrows)) return -1;
  int64_t nb = 0, nz = 0;
  int rc = dbcsr_amd_bcsr_twin_count(h, src, mode, dst.m.row_p, &nb, &nz, stream);
  if (rc) return rc;
  if (alloc_arrays(dst, src->nblkrows, src->nblkcols, src->row_blk_size, src->col_blk_size, nb, nz, elem_size(dt), false)) return -1;
  return dbcsr_amd_bcsr_twin_apply(h, dt, src, mode, antisymmetric, &dst.m, stream);
}

// sum of the block sizes of one dimension (device array of n int32): dbcsr_nfullrows_total / dbcsr_nfullcols_total
int full_extent(const int32_t* sizes, int n, int64_t* out, hipStream_t st) {
  *out = 0;
  if (n <= 0) return 0;
  int32_t* h = static_cast<int32_t*>(malloc(sizeof(int32_t) * (size_t)n));
  if (!h) return -1;
  if (hipMemcpyAsync(h, sizes, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    free(h);
    return -1;
  }
  int64_t t = 0;
  for (int i = 0; i < n; ++i) t += h[i];
  free(h);
  *out = t;
  return 0;
}

// C without any block (what dbcsr_multiply_generic makes of the product matrix when keep_product_data is false,
// src/mm/dbcsr_mm.F:865-870): only row_p exists, all zeros
int empty_like(const dbcsr_amd_bcsr* c, Owned& e, hipStream_t st) {
  if (alloc_row_p(e, c->nblkrows)) return -1;
  if (hipMemsetAsync(e.m.row_p, 0, sizeof(int32_t) * ((size_t)c->nblkrows + 1), st) != hipSuccess) return -1;
  e.m.nblkrows = c->nblkrows;
  e.m.nblkcols = c->nblkcols;
  e.m.row_blk_size = c->row_blk_size;
  e.m.col_blk_size = c->col_blk_size;
  e.m.nblks = 0;
  return 0;
}

int symbolic_numeric(void* h, libsmm_acc_data_t dt, double alpha, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, double beta,
                     const dbcsr_amd_bcsr* c_in, int retain, double eps, Owned& out, dbcsr_amd_mm_counts* counts, void* stream) {
  if (alloc_row_p(out, c_in->nblkrows)) return -1;
  int rc = dbcsr_amd_mm_symbolic_filtered(h, dt, alpha, eps, a, b, c_in, retain, out.m.row_p, counts, stream);
  if (rc) return rc;
  if (alloc_arrays(out, c_in->nblkrows, c_in->nblkcols, c_in->row_blk_size, c_in->col_blk_size, counts->c_nblks, counts->c_nze, elem_size(dt),
                   false))
    return -1;
  // (this helper's product goes straight into the block filter with the same eps: the blocks it will drop need not be written)
  if (eps > 0.0 && !retain) {
    static const bool off = getenv("DBCSR_AMD_MM_EXPECT_FILTER") && atoi(getenv("DBCSR_AMD_MM_EXPECT_FILTER")) == 0;
    if (!off) dbcsr_amd_mm_expect_filter(h, eps);
  }
  return dbcsr_amd_mm_numeric(h, dt, alpha, a, b, beta, c_in, &out.m, stream);
}

// L2 blocking over k (see MultiplyEngine.multiply_local in dbcsr_amd/multiply.py for the measurements): when A's average block
// row is larger than 1 MB neither operand stays L2-resident; the product is then formed as one symbolic product of the whole
// operands (C's final structure, C = beta*C_in on it) followed by passes over k ranges that accumulate in place.
int k_passes(void* h, libsmm_acc_data_t dt, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, double filter_eps, void* stream) {
  if (filter_eps > 0.0) return 1;  // the on-the-fly filter counts the blocks of a whole A row
  if (const char* f = getenv("DBCSR_AMD_MM_KCHUNKS")) return atoi(f) > 1 ? atoi(f) : 1;
  if (a->nblkcols < 64 || a->nblkrows < 1) return 1;
  // The probe below runs on the engine and costs it its plan (it shares the symbolic phase's work areas): a loop that multiplies the
  // same resident operand again and again (dbcsr_amd_dev_multiply) would pay a full symbolic phase per call for it -- 1.6 of config 2's
  // 20.3 ms from a Fortran host (gpurun_out/r05_s05).  The answer only steers speed, so it is remembered per engine for the operand whose
  // index arrays sit at the same addresses with the same shape, data type, block count and (non-zero) stamp.
  KPassMemo* memo = engine_kpass_memo(h);
  const bool stamped = memo && a->index_stamp != 0;  // stamp 0 = the owner does not track its index arrays: never trusted
  if (stamped && memo->row_p == a->row_p && memo->blk_p == a->blk_p && memo->nblks == (int64_t)a->nblks && memo->stamp == (uint64_t)a->index_stamp &&
      memo->dt == (int)dt && memo->nblkrows == a->nblkrows && memo->nblkcols == a->nblkcols && memo->b_nblks == (b ? (int64_t)b->nblks : -1))
    return memo->npass;
  auto remember = [&](int n) {
    if (stamped) {
      memo->row_p = a->row_p, memo->blk_p = a->blk_p, memo->nblks = (int64_t)a->nblks, memo->stamp = (uint64_t)a->index_stamp;
      memo->dt = (int)dt, memo->nblkrows = a->nblkrows, memo->nblkcols = a->nblkcols, memo->npass = n, memo->b_nblks = b ? (int64_t)b->nblks : -1;
    }
    return n;
  };
  Owned probe;
  if (alloc_row_p(probe, a->nblkrows)) return 1;
  int64_t nb = 0, nz = 0;
  if (dbcsr_amd_bcsr_crop_count(h, dt, a, -1, -1, -1, -1, probe.m.row_p, &nb, &nz, stream)) return 1;
  const double row_bytes = (double)nz * (double)elem_size(dt) / (double)a->nblkrows;
  if (row_bytes <= 1.0 * 1048576.0) return remember(1);      // (round 5: 23 x 23 at 20 % fill, rows of 1.2 MB, gains 7 % from two passes)
  if (nb > 0 && nz > 1024 * nb) return remember(1);           // blocks above 32 x 32 on average: the workgroup-per-C-block kernel shares its operands in LDS
  // (round 6, session r06_49: a pass costs C's bytes and saves the operands': it pays only when a C block collects many products -- 32^3 at 10 % fill, 14 per
  //  C block: 44.0 ms in one pass against 53.0 in two; 23^3 at 20 %, 57: 83.5 against 75.3 -- at least 48, estimated from the operands' fills)
  const double fill_a = (double)nb / ((double)a->nblkrows * (double)a->nblkcols);
  const double fill_b = b && b->nblkrows > 0 && b->nblkcols > 0 ? (double)b->nblks / ((double)b->nblkrows * (double)b->nblkcols) : fill_a;
  if ((double)a->nblkcols * fill_a * fill_b < 48.0) return remember(1);
  const int n = (int)std::ceil(row_bytes / 1048576.0);
  return remember(n > 8 ? 8 : n);
}

int multiply_in_k_passes(void* h, libsmm_acc_data_t dt, double alpha, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, double beta,
                         const dbcsr_amd_bcsr* c_in, int retain, int npass, Owned& out, dbcsr_amd_mm_counts* counts, void* stream) {
  hipStream_t st = stream_of(stream);
  if (alloc_row_p(out, c_in->nblkrows)) return -1;
  int rc = dbcsr_amd_mm_symbolic(h, a, b, c_in, retain, out.m.row_p, counts, stream);
  if (rc) return rc;
  if (alloc_arrays(out, c_in->nblkrows, c_in->nblkcols, c_in->row_blk_size, c_in->col_blk_size, counts->c_nblks, counts->c_nze, elem_size(dt),
                   false))
    return -1;
  if ((rc = dbcsr_amd_mm_init_c(h, dt, beta, c_in, &out.m, stream))) return rc;
  // k ranges on block boundaries: element offsets of A's block columns
  const int nbk = a->nblkcols;
  int32_t* ksz = static_cast<int32_t*>(malloc(sizeof(int32_t) * (size_t)nbk));
  if (!ksz) return -1;
  if (hipMemcpyAsync(ksz, a->col_blk_size, sizeof(int32_t) * (size_t)nbk, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    free(ksz);
    return -1;
  }
  int64_t* off = static_cast<int64_t*>(malloc(sizeof(int64_t) * ((size_t)nbk + 1)));
  off[0] = 0;
  for (int k = 0; k < nbk; ++k) off[k + 1] = off[k] + ksz[k];
  free(ksz);
  Owned tmp_row_p;
  if (alloc_row_p(tmp_row_p, c_in->nblkrows)) {
    free(off);
    return -1;
  }
  int64_t flop = 0, nprod = 0;
  for (int p = 0; p < npass && rc == 0; ++p) {
    const int kb0 = (int)((int64_t)p * nbk / npass), kb1 = (int)((int64_t)(p + 1) * nbk / npass);
    if (kb1 <= kb0) continue;
    Owned ac, bc;
    if ((rc = crop(h, dt, a, -1, -1, off[kb0], off[kb1] - 1, ac, nullptr, stream))) break;
    if ((rc = crop(h, dt, b, off[kb0], off[kb1] - 1, -1, -1, bc, nullptr, stream))) break;
    dbcsr_amd_mm_counts cnt;
    if ((rc = dbcsr_amd_mm_symbolic(h, &ac.m, &bc.m, &out.m, 1, tmp_row_p.m.row_p, &cnt, stream))) break;
    dbcsr_amd_bcsr acc = out.m;  // same arrays, in place
    acc.row_p = tmp_row_p.m.row_p;
    if ((rc = dbcsr_amd_mm_numeric(h, dt, alpha, &ac.m, &bc.m, 1.0, &out.m, &acc, stream))) break;
    flop += cnt.flop;
    nprod += cnt.nproducts;
    if (hipStreamSynchronize(st) != hipSuccess) rc = -1;  // the cropped operands are freed at the end of this iteration
  }
  free(off);
  counts->flop = flop;
  counts->nproducts = nprod;
  return rc;
}

}  // namespace

extern "C" {

int dbcsr_amd_bcsr_release(dbcsr_amd_bcsr* m) {
  if (!m) return -1;
  (void)pool_free(m->row_p);
  (void)pool_free(m->col_i);
  (void)pool_free(m->blk_p);
  (void)pool_free(m->data);
  m->row_p = m->col_i = nullptr;
  m->blk_p = nullptr;
  m->data = nullptr;
  m->nblks = 0;
  return 0;
}

int dbcsr_amd_multiply(void* handle, char transa, char transb, libsmm_acc_data_t datatype, double alpha, const dbcsr_amd_bcsr* matrix_a,
                       const dbcsr_amd_bcsr* matrix_b, double beta, const dbcsr_amd_bcsr* matrix_c, const int64_t* limits, int retain_sparsity,
                       double filter_eps, dbcsr_amd_bcsr* c_out, int64_t* flop, void* stream) {
  if (!handle || !matrix_a || !matrix_b || !matrix_c || !c_out) return -1;
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  auto is_n = [](char t) { return t == 'N' || t == 'n'; };
  auto is_t = [](char t) { return t == 'T' || t == 't' || t == 'C' || t == 'c'; };  // real data: 'C' == 'T'
  if ((!is_n(transa) && !is_t(transa)) || (!is_n(transb) && !is_t(transb))) {
    fprintf(stderr, "dbcsr_amd_multiply: invalid transpose flag\n");
    return -1;
  }
  hipStream_t st = stream_of(stream);
  const size_t esz = elem_size(datatype);
  int rc = 0;
  // op(A), op(B)
  Owned ta, tb;
  const dbcsr_amd_bcsr* A = matrix_a;
  const dbcsr_amd_bcsr* B = matrix_b;
  if (is_t(transa)) {
    if ((rc = transposed(handle, datatype, matrix_a, ta, stream))) return rc;
    A = &ta.m;
  }
  if (is_t(transb)) {
    if ((rc = transposed(handle, datatype, matrix_b, tb, stream))) return rc;
    B = &tb.m;
  }
  if (A->nblkcols != B->nblkrows || A->nblkrows != matrix_c->nblkrows || B->nblkcols != matrix_c->nblkcols) {
    fprintf(stderr, "dbcsr_amd_multiply: incompatible block dimensions\n");
    return -1;
  }
  // submatrix limits: 1-based inclusive full-matrix indices, 0 = not given.  Validation and the "optimise the defaults away"
  // step are the reference's (src/mm/dbcsr_mm.F:631-692): an invalid limit is an error there (DBCSR_ABORT), never clamped.
  Owned ca, cb, cc, cempty;
  const dbcsr_amd_bcsr* Cin = matrix_c;
  double beta_eff = beta;
  bool limited = false;
  if (limits)
    for (int i = 0; i < 6; ++i) limited = limited || limits[i] != 0;
  int64_t f_row = 0, l_row = 0, f_col = 0, l_col = 0, f_k = 0, l_k = 0;
  bool window_keeps = false;  // a row / column window that ends inside C: blocks outside it must survive
  if (limited) {
    int64_t nr = 0, nc = 0, nk = 0;
    if (full_extent(matrix_c->row_blk_size, matrix_c->nblkrows, &nr, st) || full_extent(matrix_c->col_blk_size, matrix_c->nblkcols, &nc, st) ||
        full_extent(A->col_blk_size, A->nblkcols, &nk, st))
      return -1;
    f_row = limits[0]; l_row = limits[1]; f_col = limits[2]; l_col = limits[3]; f_k = limits[4]; l_k = limits[5];
    if (f_row < 0 || f_row > nr || l_row < 0 || l_row > nr || f_col < 0 || f_col > nc || l_col < 0 || l_col > nc || f_k < 0 || f_k > nk || l_k < 0 ||
        l_k > nk || (l_row && f_row > l_row) || (l_col && f_col > l_col) || (l_k && f_k > l_k)) {
      fprintf(stderr, "dbcsr_amd_multiply: invalid limits (rows %lld..%lld of %lld, columns %lld..%lld of %lld, k %lld..%lld of %lld)\n",
              (long long)f_row, (long long)l_row, (long long)nr, (long long)f_col, (long long)l_col, (long long)nc, (long long)f_k, (long long)l_k,
              (long long)nk);
      return -1;
    }
    if (f_row == 1) f_row = 0;
    if (l_row == nr) l_row = 0;
    if (f_col == 1) f_col = 0;
    if (l_col == nc) l_col = 0;
    if (f_k == 1) f_k = 0;
    if (l_k == nk) l_k = 0;
    window_keeps = (l_col > 0 && l_col < nc) || (l_row > 0 && l_row < nr);
    limited = f_row || l_row || f_col || l_col || f_k || l_k;
  }
  // Product data is retained when retain_sparsity, beta != 0, or the row / column window ends inside C (dbcsr_mm.F:695-704);
  // otherwise the old C is discarded before the multiplication: its blocks disappear and their values are never read.
  const bool keep_product_data = retain_sparsity || beta != 0.0 || window_keeps;
  if (!keep_product_data) {
    if (empty_like(matrix_c, cempty, st)) return -1;
    Cin = &cempty.m;
    beta_eff = 1.0;
  }
  if (limited) {
    const int64_t r0 = f_row ? f_row - 1 : -1, r1 = l_row ? l_row - 1 : -1;
    const int64_t c0 = f_col ? f_col - 1 : -1, c1 = l_col ? l_col - 1 : -1;
    const int64_t k0 = f_k ? f_k - 1 : -1, k1 = l_k ? l_k - 1 : -1;
    if ((rc = crop(handle, datatype, A, r0, r1, k0, k1, ca, nullptr, stream))) return rc;
    if ((rc = crop(handle, datatype, B, k0, k1, c0, c1, cb, nullptr, stream))) return rc;
    A = &ca.m;
    B = &cb.m;
    if (beta != 1.0 && keep_product_data) {  // dbcsr_scale(matrix_c, beta, limits): on a copy, the caller's C stays as it is
      int64_t nz = 0;
      if ((rc = crop(handle, datatype, matrix_c, -1, -1, -1, -1, cc, &nz, stream))) return rc;
      if ((rc = dbcsr_amd_bcsr_scale_window(handle, datatype, &cc.m, beta, r0, r1, c0, c1, stream))) return rc;
      Cin = &cc.m;
    }
    beta_eff = 1.0;
  }
  // the product (with the on-the-fly filter), then the final block filter
  Owned prod;
  dbcsr_amd_mm_counts counts;
  const int npass = k_passes(handle, datatype, A, B, filter_eps, stream);
  if (npass > 1) {
    if ((rc = multiply_in_k_passes(handle, datatype, alpha, A, B, beta_eff, Cin, retain_sparsity, npass, prod, &counts, stream))) return rc;
  } else if ((rc = symbolic_numeric(handle, datatype, alpha, A, B, beta_eff, Cin, retain_sparsity, filter_eps, prod, &counts, stream))) {
    return rc;
  }
  if (flop) *flop = counts.flop;
  Owned* result = &prod;
  Owned filtered;
  if (filter_eps > 0.0 && !retain_sparsity) {
    if (alloc_row_p(filtered, prod.m.nblkrows)) return -1;
    int64_t nb = 0, nz = 0;
    if ((rc = dbcsr_amd_bcsr_filter_count(handle, datatype, &prod.m, filter_eps, filtered.m.row_p, &nb, &nz, stream))) return rc;
    if (nb != prod.m.nblks) {  // (nothing below the threshold: the product is the result, no second copy)
      if (alloc_arrays(filtered, prod.m.nblkrows, prod.m.nblkcols, prod.m.row_blk_size, prod.m.col_blk_size, nb, nz, esz, false)) return -1;
      if ((rc = dbcsr_amd_bcsr_filter_apply(handle, datatype, &prod.m, &filtered.m, stream))) return rc;
      result = &filtered;
    }
  }
  // temporaries are freed when this function returns: everything that reads them must have finished
  if (hipStreamSynchronize(st) != hipSuccess) return -1;
  *c_out = result->m;
  c_out->row_blk_size = matrix_c->row_blk_size;
  c_out->col_blk_size = matrix_c->col_blk_size;
  result->live = false;  // ownership passes to the caller (dbcsr_amd_bcsr_release)
  return 0;
}

int dbcsr_amd_bcsr_desymmetrized(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, int antisymmetric, dbcsr_amd_bcsr* dst,
                                 void* stream) {
  if (!handle || !src || !dst) return -1;
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  Owned full;
  int rc = twin(handle, datatype, src, 0, antisymmetric, full, stream);
  if (rc == 0 && hipStreamSynchronize(stream_of(stream)) != hipSuccess) rc = -1;
  if (rc) return rc;
  *dst = full.m;
  full.live = false;  // the caller releases it (dbcsr_amd_bcsr_release)
  return 0;
}

int dbcsr_amd_multiply_symmetric_c(void* handle, char transa, char transb, libsmm_acc_data_t datatype, double alpha,
                                   const dbcsr_amd_bcsr* matrix_a, const dbcsr_amd_bcsr* matrix_b, double beta, const dbcsr_amd_bcsr* matrix_c,
                                   int antisymmetric, int retain_sparsity, double filter_eps, dbcsr_amd_bcsr* c_out, int64_t* flop, void* stream) {
  return dbcsr_amd_multiply_symmetric_c_klimits(handle, transa, transb, datatype, alpha, matrix_a, matrix_b, beta, matrix_c, antisymmetric, 0, 0,
                                                retain_sparsity, filter_eps, c_out, flop, stream);
}

// ... with limits on the INNER dimension (first_k / last_k of dbcsr_multiply: 1-based inclusive element indices, 0 = not given).  They crop op(A)'s
// columns and op(B)'s rows and have nothing to do with C's symmetry; the reference's own tests multiply into symmetric products with exactly these
// (tests/dbcsr_test_multiply.F:196-200: full row / column limits, any k limits).  Row / column limits together with a product with symmetry are not
// offered (the caller leaves such a multiply to the reference path).
int dbcsr_amd_multiply_symmetric_c_klimits(void* handle, char transa, char transb, libsmm_acc_data_t datatype, double alpha,
                                           const dbcsr_amd_bcsr* matrix_a, const dbcsr_amd_bcsr* matrix_b, double beta, const dbcsr_amd_bcsr* matrix_c,
                                           int antisymmetric, int64_t first_k, int64_t last_k, int retain_sparsity, double filter_eps,
                                           dbcsr_amd_bcsr* c_out, int64_t* flop, void* stream) {
  if (!handle || !matrix_c || !c_out || matrix_c->nblkrows != matrix_c->nblkcols) return -1;
  const int64_t limits[6] = {0, 0, 0, 0, first_k, last_k};
  if (datatype != dbcsr_type_real_8 && datatype != dbcsr_type_real_4) return -10;
  // dbcsr_mm.F:711-719: the index of a product matrix with symmetry is put into canonical form before the multiplication ...
  Owned canon;
  int rc = twin(handle, datatype, matrix_c, 1, antisymmetric, canon, stream);
  if (rc) return rc;
  // ... the local multiply computes only the blocks stored in that form (dbcsr_mm_csr.F:280-292) ...
  dbcsr_amd_bcsr prod;
  if ((rc = dbcsr_amd_mm_set_canonical_product(handle, 1))) return rc;
  rc = dbcsr_amd_multiply(handle, transa, transb, datatype, alpha, matrix_a, matrix_b, beta, &canon.m, (first_k || last_k) ? limits : nullptr,
                          retain_sparsity, filter_eps, &prod, flop, stream);
  (void)dbcsr_amd_mm_set_canonical_product(handle, 0);
  if (rc) return rc;
  // ... and the result goes back to the stored triangle (row <= column)
  Owned upper;
  rc = twin(handle, datatype, &prod, 2, antisymmetric, upper, stream);
  if (rc == 0 && hipStreamSynchronize(stream_of(stream)) != hipSuccess) rc = -1;
  (void)dbcsr_amd_bcsr_release(&prod);
  if (rc) return rc;
  *c_out = upper.m;
  c_out->row_blk_size = matrix_c->row_blk_size;
  c_out->col_blk_size = matrix_c->col_blk_size;
  upper.live = false;
  return 0;
}

}  // extern "C"
