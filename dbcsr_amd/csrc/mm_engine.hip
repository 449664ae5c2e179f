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
//   mm_numeric_f64_small<D>                                every block dimension at most 8: one 8 x 8 tile per wave, whole blocks per 8-byte load (mm_numeric_f64_small.h)
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

#include "mm_workspace.h"
#include "mm_symbolic.h"
#include "mm_numeric_f64.h"
#include "mm_numeric_f64_big.h"
#include "mm_numeric_f64_small.h"
#include "mm_mid.h"   // the one-wave slab kernels of the blocks of 25 ... 40 (mm_numeric_f64_mid.h, mm_mid.hip)
#include "mm_numeric_f32.h"
#include "mm_aux.h"
// The library comes in two builds (Makefile): the SHIPPING one holds what a multiply can run by itself -- the kernels listed above, their
// symbolic phases, plan reuse -- and the LAB one (-DDBCSR_AMD_EXPERIMENTS, libdbcsr_acc_amd_lab.so) adds every dataflow and variant that
// was built, made parity-green and measured but does not win: the LDS-DMA ring kernels (mm_dma.h), XCD-wide C tiles (mm_tile.*), CU-wide
// C tiles with B shared in LDS (mm_band.*), the persistent form and the ablation / keep-alive variants of the exact-size kernel, the
// G-block bodies and stream spreading of the class kernels, the occupancy and row-group knobs.  Their switches exist in the lab build only.
#ifdef DBCSR_AMD_EXPERIMENTS
#include "mm_lab_api.h"
#include "mm_group.h"
#include "mm_group64.h"
#include "mm_dma.h"
#include "mm_tile_index.h"
#include "mm_band_index.h"
#endif
namespace dbcsr_amd {

#include "mm_engine_state.h"    // plan_compare, struct Engine, helpers
#include "mm_engine_env.h"      // engine_read_env: the environment switches, read once per engine
#include "mm_engine_launch.h"   // kernel tables and launch dispatchers
#include "mm_engine_plan.h"     // plan reuse
#ifdef DBCSR_AMD_EXPERIMENTS
#include "mm_engine_lab.h"      // host side of the experimental dataflows
#endif

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
  if (hipHostMalloc(reinterpret_cast<void**>(&E->cls_host_hist), 3 * 33 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&E->cls_host_lens), (2 * kNumClasses + 1) * sizeof(int64_t), hipHostMallocDefault) != hipSuccess)
    return -1;
  if (hipHostMalloc(reinterpret_cast<void**>(&E->plan_host_flag), sizeof(int), hipHostMallocDefault) != hipSuccess) return -1;
  engine_read_env(E);
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
  E->hot_counters.release();
  E->a_bm.release(); E->bt_bm.release(); E->tile_prog.release(); E->a_pre.release(); E->tile_rows.release(); E->tile_cols.release();
  E->tile_cnt.release(); E->tile_flags.release(); E->tile_start.release();
#ifdef DBCSR_AMD_EXPERIMENTS
  E->tdescs.release(); E->tentries.release(); E->band_descs_buf.release(); E->band_entries.release(); E->band_rem.release();
#endif
  E->band_cnt_list.release(); E->band_cnt_b.release();
  E->band_cnt_rem.release(); E->band_sub_cnt.release(); E->band_flags.release(); E->band_list_off.release(); E->band_seq_off.release();
  E->band_rem_start.release(); E->band_times.release(); E->band_prog.release();
  if (E->host_scalars) (void)hipHostFree(E->host_scalars);
  if (E->plan_host_flag) (void)hipHostFree(E->plan_host_flag);
  E->plan_words.release(); E->plan_c_col_i.release(); E->plan_c_blk_p.release(); E->plan_flag.release(); E->work.release();
  if (E->cls_host_hist) (void)hipHostFree(E->cls_host_hist);
  if (E->cls_host_lens) (void)hipHostFree(E->cls_host_lens);
  E->cls_hist.release(); E->cls_row.release(); E->cls_col.release(); E->cls_col_bm.release(); E->cls_lens.release(); E->cls_vpos.release(); E->cls_vrow.release();
  for (int i = 0; i < 3; ++i)
    if (E->ev[i]) (void)hipEventDestroy(E->ev[i]);
  for (int i = 0; i < 3; ++i) {
    if (E->side_stream[i]) (void)hipStreamDestroy(E->side_stream[i]);
    if (E->join_ev[i]) (void)hipEventDestroy(E->join_ev[i]);
  }
  if (E->fork_ev) (void)hipEventDestroy(E->fork_ev);
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
  E->drop_pending = 0.0;   // (an announced final filter belongs to ONE numeric phase: a new symbolic phase cancels whatever an abandoned multiply left)
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
  // block-size maxima (LDS slice size / kernel choice of the numeric phase), the most frequent block size per dimension (choice of an exact-size kernel), the most
  // frequent size in units of 4 of C's rows and columns (the slab kernels' exact launch when no size dominates) and the histograms of the sizes 1 ... 32 (the
  // (m, n) classes of a mixed-size multiply): one launch (block_size_stats)
  E->cls_mode = false;
  if (E->use_classes > 0 && E->cls_hist.ensure(3 * 33)) return -1;
  hipLaunchKernelGGL(block_size_stats, dim3(3), dim3(256), 0, st, a->row_blk_size, nbr, a->col_blk_size, nbk, b->col_blk_size, nbc,
                     reinterpret_cast<int*>(E->dev_scalars.p + 4), reinterpret_cast<int*>(E->dev_scalars.p + 8), reinterpret_cast<int*>(E->dev_scalars.p + 11),
                     E->use_classes > 0 ? E->cls_hist.p : (int*)nullptr);
  if (E->use_classes > 0) ACC_CHECK(hipMemcpyAsync(E->cls_host_hist, E->cls_hist.p, 3 * 33 * sizeof(int), hipMemcpyDeviceToHost, st));
  // need c_nblks (and the block-size extrema) on the host to size per-block work arrays
  ACC_CHECK(hipMemcpyAsync(E->host_scalars, dsc, 13 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
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
    const int* um = reinterpret_cast<const int*>(E->host_scalars + 11);
    E->units_m = um[0], E->units_cnt_m = um[1], E->units_n = um[2], E->units_cnt_n = um[3];
  }
  // (m, n) classes: blocks of at most 32 in every dimension, no single dominant size (that case has its ahead-of-time
  // kernel), not the packed 4 x 4 case, and enough C blocks to pay for compiling the class kernels (forced with
  // DBCSR_AMD_MM_CLASSES=2)
  // (a dominant triplet that is NOT a cube of 9 ... 32 has no ahead-of-time kernel: uniform rectangular blocks -- 5 x 13 x 23, 23 x 23 x 5 -- took the
  //  run-time-size kernel until round 6, session 43; they are one class with one inner size)
  const bool hot_cube = E->hot_m >= 9 && E->hot_m == E->hot_n && E->hot_m == E->hot_k;
  if (E->use_classes > 0 && E->max_m <= 32 && E->max_k <= 32 && E->max_n <= 32 && E->min_m >= 1 && E->min_k >= 1 && E->min_n >= 1 &&
      !(E->max_m <= 4 && E->max_n <= 4) && (E->use_classes > 1 || (!hot_cube && c_nblks >= 200000))) {
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
        E->cls_lens.ensure(2 * kNumClasses + 1) || E->cls_vpos.ensure((size_t)nbr + 1) || E->cls_vrow.ensure((size_t)nbr + 1))
      return -1;
    hipLaunchKernelGGL(class_ids, grid_for(nbr), dim3(256), 0, st, a->row_blk_size, nbr, E->cls_m[0], E->cls_m[1], E->cls_m[2], E->cls_row.p);
    hipLaunchKernelGGL(class_ids, grid_for(nbc), dim3(256), 0, st, b->col_blk_size, nbc, E->cls_n[0], E->cls_n[1], E->cls_n[2], E->cls_col.p);
    hipLaunchKernelGGL(class_col_bitmaps, grid_for(W), dim3(256), 0, st, E->cls_col.p, nbc, W, E->cls_col_bm.p);
    hipLaunchKernelGGL(class_row_deal, dim3(1), dim3(256), 0, st, E->cls_row.p, nbr, E->cls_vpos.p, E->cls_vrow.p);
    hipLaunchKernelGGL(order_count_cls, grid_for(nkeys), dim3(256), 0, st, E->c_bm.p, E->cls_row.p, E->cls_col_bm.p, E->cls_vrow.p, nbr, W, PW, NP, R,
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
                         E->cls_col_bm.p, E->order_base.p, E->cls_lens.p, E->cls_vpos.p, nbr, W, PW, NP, R, E->order.p);
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
  if (!reuse) E->work_built = E->tile_built = E->band_built = E->group_built = false;
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
    const int nJr = (b->nblkcols + 63) / 64;
    hipLaunchKernelGGL(finish_descs_grid, grid_for((int64_t)nbr * nJr * 64), dim3(256), 0, st, c_in->row_p, c_in->blk_p, c_out->row_blk_size,
                       c_out->col_blk_size, E->have_cin ? E->cin_bm.p : (const uint32_t*)nullptr,
                       E->have_cin ? E->cin_pre.p : (const int*)nullptr, E->c_bm.p, E->c_pre.p, c_out->row_p, E->c_blk_p_ws.p, E->prod_start.p,
                       E->prod_cnt.p, nbr, W, nJr, c_out->col_i, c_out->blk_p, E->descs.p);
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
  // the one-wave slab kernel (mm_numeric_f64_mid.h): fp64 C blocks whose dominant (else largest) size has a dimension of 33 ... 40 and the other of
  // 21 ... 40 (mid_f64_serves), any inner dimension; its second launch takes the blocks of another size.  Mixed-size multiplies (cls_mode) ask per class below.
  int mid_rb = 0, mid_cb = 0;
  if (datatype == dbcsr_type_real_8 && E->use_big && E->use_mid && E->use_lds && !E->cls_mode && E->max_m <= 48 && E->max_n <= 48 && E->min_m >= 1 &&
      E->min_n >= 1 && E->min_k >= 1 && E->order_len > 0 && !(E->dbg & ~32) && !E->dma_stages && !E->hot_persistent && E->hot_variant == 0 && E->use_hot &&
      E->use_pipe != 1) {
    // (without a dominant size -- the size statistics stop at 32 -- the largest size is multiplied exactly when the blocks go beyond 32, where the
    // alternative is the workgroup kernel, or when every block is in the range: the second launch pads the others to 40 x 40)
    const bool dom = E->hot_m > 0 && E->hot_n > 0, all_in = (E->min_m > 24 && E->min_n > 24) || E->max_m > 32 || E->max_n > 32;
    const int dm = dom ? E->hot_m : (all_in ? E->max_m : 0), dn = dom ? E->hot_n : (all_in ? E->max_n : 0);
    if (dm > 0 && dn > 0 && mid_f64_serves(dm, dn, 0)) mid_rb = (dm + 3) / 4, mid_cb = (dn + 3) / 4;
    if (mid_rb && !dom) {
      // No dominant size: the exact launch would serve a minority and the second launch -- the largest shape, 10 x 10 or 12 x 12 units -- pads everything
      // else (30 / 36 mixed: 43 ms against 29 through the workgroup kernel, session r06_47).  The slab kernel stays when ONE launch serves every block -- the
      // largest blocks ARE the largest shape (30 / 40, 34 / 40, 23 / 40: +14-17 %) -- or when at least 80 % of C's rows and of its columns have the exact
      // launch's units (33 / 36: +17 %; 36 with a tail block).
      const int mu = (std::max(E->max_m, E->max_n) + 3) / 4, fb = (mu > 10 || mid_rb > 10 || mid_cb > 10) ? 12 : 10;
      const bool single = mid_rb == fb && mid_cb == fb;
      const bool most = 10ll * E->units_cnt_m >= 8ll * nbr && 10ll * E->units_cnt_n >= 8ll * b->nblkcols && mid_f64_serves(4 * E->units_m, 4 * E->units_n, 0);
      if (most)
        mid_rb = E->units_m, mid_cb = E->units_n;   // (the exact launch takes the most frequent shape, the second launch the rest)
      else if (!single)
        mid_rb = mid_cb = 0;
    }
  }
  // every block dimension at most 8 (and not the packed 4 x 4 case): one 8 x 8 tile per wave, several products in flight (mm_numeric_f64_small.h)
  const bool tiny4 = E->use_tiny && E->max_m <= 4 && E->max_n <= 4;
  const bool small8 = datatype == dbcsr_type_real_8 && E->use_small > 0 && E->use_lds && !tiny4 && E->max_m <= 8 && E->max_n <= 8 && E->max_k <= 8 && E->min_m >= 1 &&
                      E->min_n >= 1 && E->min_k >= 1 && !(E->dbg & ~32) && !E->dma_stages && !E->hot_persistent && E->hot_variant == 0 && E->use_pipe < 0;  // (DBCSR_AMD_MM_KERNEL=lds1 | pipe ask for those kernels)
  const Work* hot_work = nullptr;
  {
    const bool small64 = datatype == dbcsr_type_real_8 && E->use_lds && E->max_m <= 32 && E->max_k <= 32 && E->max_n <= 32 && E->min_m >= 1 &&
                         E->min_k >= 1 && E->min_n >= 1 && !tiny4 && !small8;
    // (the ahead-of-time exact-size kernel reads nothing else; the class kernels keep the order[] -> descs[] path for DBCSR_AMD_MM_WORK=0)
    const bool exact = E->cls_mode ? (E->class_g == 1 && E->use_work)
                                   : (E->use_hot && E->use_pipe != 1 && E->hot_m > 0 && E->dma_stages == 0 && E->hot_m == E->hot_n && E->hot_m == E->hot_k);
    const int64_t npos = 8 * E->order_len;
    if (((small64 && exact) || mid_rb || (small8 && E->use_work)) && npos > 0) {
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
  if ((hot_work || (E->cls_mode && E->class_g == 1)) && !small8 && datatype == dbcsr_type_real_8 && E->filter.a_norms && !skip_empty && !E->retain) {
    if (E->norms64.ensure((size_t)nblk + 1)) return -1;
    epi_norms = E->norms64.p;
    if (E->dbg & 8) epi_norms = nullptr;  // (profiling epilogue of the exact-size kernel: it leaves no norms, the filter then computes them)
  }
  // the final block filter announced for this numeric phase (dbcsr_amd_mm_expect_filter): its eps^2 rides behind the norms, norms64[nblk], where the exact-size and
  // class kernels pick it up -- a block below it is not written
  {
    const double drop = epi_norms ? E->drop_pending : 0.0;
    E->drop_pending = 0.0;
    E->unwritten_below = 0.0;
    if (epi_norms) {
      hipLaunchKernelGGL(store_scalar_f64, dim3(1), dim3(1), 0, st, epi_norms + nblk, drop);
      E->unwritten_below = drop;
    }
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
      const double* ad = static_cast<const double*>(a->data);
      const double* bd = static_cast<const double*>(b->data);
      double* cd = static_cast<double*>(c_out->data);
      const double* cid = static_cast<const double*>(c_in->data);
      if (nwg_t > 0) {
        auto tiny = E->max_k > 4 ? mm_numeric_f64_tiny<false> : mm_numeric_f64_tiny<true>;
        hipLaunchKernelGGL(tiny, dim3(nwg_t), dim3(256), 0, st, E->descs.p, nblk, E->entries.p, ad, bd, cd, cid, alpha, beta, skip_empty, E->order.p);
      }
    } else if (small8) {
      // C blocks per wave (DBCSR_AMD_MM_SMALL_G; 0 = by the list length): with one or two products per C block a wave lives for a microsecond and the launch is
      // bound by the rate at which waves start (5 x 5 blocks at 1 % fill, 14 M C blocks: 4.96 ms with one block per wave, 4.2 with eight); with fourteen it is not
      const int sg = E->small_group > 0 ? E->small_group : (E->nproducts < 4 * nblk ? 8 : 1);
      const int64_t npos_s = 8 * E->order_len;
      const unsigned nwg_s = (unsigned)((npos_s + 4 * (int64_t)sg - 1) / (4 * (int64_t)sg));
      const int depth = E->use_small == 3 || E->use_small == 4 || E->use_small == 6 || E->use_small == 8 ? E->use_small : 2;
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_small<%d>", depth);
      if (nwg_s > 0) {
        auto kern = hot_work ? (depth == 2 ? mm_numeric_f64_small<2, true> : depth == 3 ? mm_numeric_f64_small<3, true> : depth == 4 ? mm_numeric_f64_small<4, true> :
                                depth == 6 ? mm_numeric_f64_small<6, true> : mm_numeric_f64_small<8, true>)
                             : (depth == 2 ? mm_numeric_f64_small<2, false> : depth == 3 ? mm_numeric_f64_small<3, false> : depth == 4 ? mm_numeric_f64_small<4, false> :
                                depth == 6 ? mm_numeric_f64_small<6, false> : mm_numeric_f64_small<8, false>);
        hipLaunchKernelGGL(kern, dim3(nwg_s), dim3(256), 0, st, E->descs.p, nblk, E->entries.p, static_cast<const double*>(a->data),
                           static_cast<const double*>(b->data), static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta,
                           skip_empty, E->order.p, hot_work, sg, npos_s);
      }
    } else if (mid_rb && launch_mid_f64(mid_rb, mid_cb, E->min_m != E->max_m || E->min_n != E->max_n, (unsigned)(8 * E->order_len), st, E->descs.p, nblk,
                                         E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                                         static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, skip_empty, E->order.p,
                                         hot_work, (std::max(E->max_m, E->max_n) + 3) / 4, epi_norms)) {
      // blocks of 25 ... 40 in both dimensions: one wave per C block, operands in slabs (mm_numeric_f64_mid.h); the dominant size (else the largest)
      // multiplied exactly, the blocks of another size by the second launch.  Every block leaves its norm to a filtered multiply (round 6, session 56).
      if (epi_norms) {
        E->norms_data = c_out->data;
        E->norms_nblks = nblk;
      }
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_mid<%d,%d>", mid_rb, mid_cb);
    } else if (small && E->use_lds && E->cls_mode) {
      // one launch per (m, n) class on its segment of order[]: the run-time compiled exact-size kernel of the class
      // (mm_exact.h, mm_jit.hip), the generic LDS kernel for class 9 (other sizes) and for classes hiprtc could not serve
      const int g_lds_a = (E->max_m * ((E->max_k + 3) & ~3) + 1) & ~1, g_lds_b = ((E->max_k * E->max_n + 127) / 128) * 128;
      const int g_lds_wave = g_lds_a + g_lds_b;
      const int g_maxt = (std::max(E->max_m, E->max_n) + 7) / 8;
      const int dbgv = E->dbg | (skip_empty ? 32 : 0);
      int njit = 0, ngen = 0, nmid = 0, jit_mask = 0;
      const hipStream_t st_main = st;
      int nside = 0, nlaunch = 0;
      if (E->class_streams > 1) {
        nside = E->class_streams - 1;
        if (!E->fork_ev) ACC_CHECK(hipEventCreateWithFlags(&E->fork_ev, hipEventDisableTiming));
        ACC_CHECK(hipEventRecord(E->fork_ev, st_main));
        for (int i = 0; i < nside; ++i) {
          if (!E->side_stream[i]) {
            ACC_CHECK(hipStreamCreateWithFlags(&E->side_stream[i], hipStreamNonBlocking));
            ACC_CHECK(hipEventCreateWithFlags(&E->join_ev[i], hipEventDisableTiming));
          }
          ACC_CHECK(hipStreamWaitEvent(E->side_stream[i], E->fork_ev, 0));
        }
      }
      for (int c = 0; c < kNumClasses; ++c) {
        if (E->cls_len[c] == 0) continue;
        {
          const int slot = nlaunch++ % (nside + 1);
          st = slot == 0 ? st_main : E->side_stream[slot - 1];
        }
        const int* ord = E->order.p + E->cls_off[c];
        const unsigned nwg_c = (unsigned)(8 * E->cls_len[c] / 4);
        ClassKernel ck;
        const int cm = c < 9 ? E->cls_m[c / 3] : 0, cn = c < 9 ? E->cls_n[c % 3] : 0;
        // classes of 29 ... 32 rows and columns, or 21 ... 24 in one of them (DBCSR_AMD_MM_MID=3: not those): the one-wave slab kernel -- half the LDS of
        // the class kernel, which stages whole blocks (17.9 KB per wave for (32, 32), 15 KB for (32, 23): two waves per SIMD)
        if (c < 9 && E->use_mid && E->use_big && E->class_g == 1 && mid_f64_serves(cm, cn, E->use_mid == 3 ? 3 : 1) &&
            launch_mid_f64((cm + 3) / 4, (cn + 3) / 4, false, (unsigned)(8 * E->cls_len[c]), st, E->descs.p, nblk, E->entries.p,
                           static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),
                           static_cast<const double*>(c_in->data), alpha, beta, skip_empty, ord, hot_work ? hot_work + E->cls_off[c] : nullptr,
                           (std::max(cm, cn) + 3) / 4, epi_norms)) {
          ++nmid;
          jit_mask |= 1 << c;   // (the class left its norms, as the run-time compiled kernels do: block_norms_unserved_classes passes it by)
        } else if (c < 9 && cm > 0 && cn > 0 && jit_class_kernel(cm, cn, E->cls_k[0], E->cls_k[1], E->cls_k[2], E->class_g, &ck) == 0) {
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
      st = st_main;
      for (int i = 0; i < nside; ++i) {
        ACC_CHECK(hipEventRecord(E->join_ev[i], E->side_stream[i]));
        ACC_CHECK(hipStreamWaitEvent(st_main, E->join_ev[i], 0));
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
      if (nmid > 0)
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_class[%d jit + %d slab + %d generic launches; m {%d,%d,%d} n {%d,%d,%d} k {%d,%d,%d}]",
                 njit, nmid, ngen, E->cls_m[0], E->cls_m[1], E->cls_m[2], E->cls_n[0], E->cls_n[1], E->cls_n[2], E->cls_k[0], E->cls_k[1], E->cls_k[2]);
      else
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_class[%d jit + %d generic launches; m {%d,%d,%d} n {%d,%d,%d} k {%d,%d,%d}]", njit,
               ngen, E->cls_m[0], E->cls_m[1], E->cls_m[2], E->cls_n[0], E->cls_n[1], E->cls_n[2], E->cls_k[0], E->cls_k[1], E->cls_k[2]);
    } else if (small && E->use_lds) {
      // per-wave LDS slice.  Staging writes whole 1 KiB chunks (128 doubles), A's chunks first, then B's: the B part may
      // start right after A's (zero-padded) block -- the tail of A's last chunk is simply overwritten by B's first chunk
      // (one wave, in-order LDS queue) -- and only B's part is rounded up to whole chunks.  For 23x23 blocks this is
      // 9.5 KB per wave instead of 10 KB, which is what lets a 4th workgroup (16 waves) fit the CU's 160 KB.
      int lds_a = (E->max_m * ((E->max_k + 3) & ~3) + 1) & ~1, lds_b = ((E->max_k * E->max_n + 127) / 128) * 128;
      if (E->hot_m > 0 && E->hot_m == E->hot_n && E->hot_m == E->hot_k && E->hot_m % 8 == 0) {
        // the exact-size kernel stages columns of 16 / 32 doubles (B: 24 too) with a pitch of + 2 (mm_numeric_f64.h: cblock_f64_exact): its A image has
        // hot_m + 2 rows per column, its B image 16 more bytes per column (128 per KiB piece at most)
        const int S = E->hot_m, cb = (S * S * 8 + 1023) / 1024;
        if (S % 16 == 0) lds_a = std::max(lds_a, (S + 2) * S);
        lds_b = std::max(lds_b, cb * (1024 + 128) / 8 + 2);
      }
      const int lds_wave = lds_a + lds_b;
      const int maxt = (std::max(E->max_m, E->max_n) + 7) / 8;
      const size_t lds_bytes = (size_t)4 * lds_wave * sizeof(double) + (size_t)E->lds_pad;
      const unsigned nwg_o = (unsigned)(8 * E->order_len / 4);
#define DBCSR_LAUNCH(T_)                                                                                                        \
  hipLaunchKernelGGL(mm_numeric_f64_lds<T_>, dim3(nwg_o * 4u / (unsigned)ww), dim3(64 * ww),                     \
                     (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad, st, E->descs.p, nblk, E->entries.p,     \
                     static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data), \
                     static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, E->dbg | (skip_empty ? 32 : 0), E->order.p)
      int tile_rc = 1;
#ifdef DBCSR_AMD_EXPERIMENTS
      // a wave per R C blocks of one block column, B shared inside the wave (mm_group64.h): one dominant cube size the kernel is built for, no
      // block norms to leave behind (filtered multiplies), no symmetric product
      if (E->f64_group >= 2 && hot_work && E->use_hot && E->use_pipe != 1 && E->dma_stages == 0 && E->hot_m == E->hot_n && E->hot_m == E->hot_k && !epi_norms &&
          !E->canonical_c && !(E->dbg & ~32) && !E->hot_persistent) {
        tile_rc = run_group_f64(E, E->f64_group, reuse, st, a, b, c_in, c_out, alpha, beta, skip_empty);
        if (tile_rc < 0) return -1;
        if (tile_rc == 0) tile_rc = 3;
      }
      // XCD-wide C tiles (mm_tile.h): one dominant cube size the tile kernel is built for, a C dense enough that sub-tiles of
      // 3 x 3 blocks have long product lists, no on-the-fly filter, no in-place accumulation, no symmetric product
      if (tile_rc != 3 && E->use_tile > 0 && hot_work && E->use_hot && E->use_pipe != 1 && E->dma_stages == 0 && E->hot_m == 23 && E->hot_n == 23 && E->hot_k == 23 &&
          !E->filter.a_norms && !skip_empty && !E->canonical_c && !epi_norms && !(E->dbg & ~32) &&
          (E->use_tile > 1 || (E->nproducts >= 8 * nblk && nblk >= 200000)))
        tile_rc = run_tile_f64<23>(E, st, a, b, c_in, c_out, alpha, beta);
      if (tile_rc < 0) return -1;
      // CU-wide C tiles, B shared in LDS (mm_band.h): the same conditions, and no retain_sparsity (its lists take C's pattern from the operands)
      if (tile_rc != 0 && tile_rc != 3 && E->use_band > 0 && hot_work && E->use_hot && E->use_pipe != 1 && E->dma_stages == 0 && E->hot_m == 23 && E->hot_n == 23 &&
          E->hot_k == 23 && !E->filter.a_norms && !skip_empty && !E->canonical_c && !epi_norms && !(E->dbg & ~32) && !E->retain && !E->hot_persistent &&
          (E->use_band > 1 || (E->nproducts >= 8 * nblk && nblk >= 200000))) {
        tile_rc = run_band_f64<23>(E, st, a, b, c_in, c_out, alpha, beta);
        if (tile_rc < 0) return -1;
        if (tile_rc == 0) tile_rc = 2;
      }
      // measured: the pipelined kernel wins when C blocks have few products (config 3: 3.7 per block, 10.4 vs 11.8 ms) and
      // loses when they have many (config 2: 14.4 per block, 32 vs 22 ms)
#endif
      if (tile_rc == 3) {
        // the C blocks of other sizes (tail block row / column): the one-wave-per-block kernel, told to leave the dominant size alone
        if (E->hot_cnt_m < nbr || E->hot_cnt_n < b->nblkcols)
          launch_hot_f64(E->hot_m, E->hot_n, E->hot_k, dim3((unsigned)(8 * E->order_len / ww)), (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad,
                         st, E->descs.p, nblk, E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                         static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, 64 | (skip_empty ? 32 : 0),
                         E->order.p, hot_work, ww, nullptr, 0);
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_group<%d,%d,%d;%d>", E->hot_m, E->hot_n, E->hot_k, E->group_R);
      } else if (tile_rc == 0 || tile_rc == 2) {
        // the tile / band kernel computed the C blocks of the dominant size (products with inner blocks of another size included);
        // this launch: the exact-size kernel over the blocks of the other sizes only
        launch_hot_f64(E->hot_m, E->hot_n, E->hot_k, dim3((unsigned)(8 * E->order_len / ww)), (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad,
                       st, E->descs.p, nblk, E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                       static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, 64, E->order.p,
                       hot_work, ww, nullptr, 0);
        snprintf(E->last_kernel, sizeof E->last_kernel, tile_rc == 2 ? "mm_numeric_f64_band<%d,%d,%d>" : "mm_numeric_f64_tile<%d,%d,%d>", E->hot_m, E->hot_n,
                 E->hot_k);
#ifdef DBCSR_AMD_EXPERIMENTS
      } else if (E->use_hot && E->use_pipe != 1 && E->hot_m > 0 && E->dma_stages > 0 &&
          launch_dma_f64(E->dma_stages, E->hot_m, E->hot_n, E->hot_k, (unsigned)(8 * E->order_len), st, E->descs.p, nblk, E->entries.p,
                         static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),
                         static_cast<const double*>(c_in->data), alpha, beta, skip_empty, E->order.p)) {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_dma<%d,%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k, E->dma_stages);
      } else if (E->hot_persistent && E->use_hot && E->use_pipe != 1 && E->hot_m == 23 && E->hot_n == 23 && E->hot_k == 23 && hot_work &&
                 !(E->dbg & ~32) && E->hot_counters.ensure(8 * 32) == 0) {
        // persistent waves, one counter per XCD (an experiment: see the kernel); 16 one-wave workgroups per CU is what the LDS slice allows
        static int n_cu_p = 0;
        if (n_cu_p == 0) {
          int dev = 0;
          ACC_CHECK(hipGetDevice(&dev));
          ACC_CHECK(hipDeviceGetAttribute(&n_cu_p, hipDeviceAttributeMultiprocessorCount, dev));
        }
        const int per_cu = std::max(1, (int)((160 * 1024) / ((size_t)lds_wave * sizeof(double) + (size_t)E->lds_pad)));
        ACC_CHECK(hipMemsetAsync(E->hot_counters.p, 0, 8 * 32 * sizeof(unsigned), st));
        hipLaunchKernelGGL((mm_numeric_f64_hot_persistent<23, 23, 23>), dim3((unsigned)(n_cu_p * per_cu)), dim3(64),
                           (size_t)lds_wave * sizeof(double) + (size_t)E->lds_pad, st, E->entries.p, static_cast<const double*>(a->data),
                           static_cast<const double*>(b->data), static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta,
                           lds_a, skip_empty ? 32 : 0, hot_work, (long)E->order_len, E->hot_counters.p, E->hot_xcd_mask, epi_norms);
        if (epi_norms) {
          hipLaunchKernelGGL(block_norms_other_sizes, grid_for(nblk * 64), dim3(256), 0, st, E->descs.p, nblk, static_cast<const double*>(c_out->data),
                             E->hot_m, E->hot_n, epi_norms);
          E->norms_data = c_out->data;
          E->norms_nblks = nblk;
        }
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_hot_persistent<%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k);
#endif
      } else if (E->use_hot && E->use_pipe != 1 && E->hot_m > 0 &&
          launch_hot_f64(E->hot_m, E->hot_n, E->hot_k, dim3((unsigned)(8 * E->order_len / ww)),
                         (size_t)ww * lds_wave * sizeof(double) + (size_t)E->lds_pad, st, E->descs.p, nblk, E->entries.p,
                         static_cast<const double*>(a->data), static_cast<const double*>(b->data), static_cast<double*>(c_out->data),
                         static_cast<const double*>(c_in->data), alpha, beta, lds_a, lds_wave, E->dbg | (skip_empty ? 32 : 0), E->order.p,
                         hot_work, ww, epi_norms, (E->dbg & ~32) ? 1 : E->hot_variant)) {
        // launched: C blocks of the dominant size take the exact-size path, the others the generic one; both leave their norms
        if (epi_norms) {
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
    } else if (E->use_big && E->use_lds && E->max_m <= 80 && E->max_n <= 80 && E->min_m >= 1 && E->min_n >= 1 && E->min_k >= 1 && !E->cls_mode &&
               E->order_len > 0 && (E->max_m > 32 || E->max_n > 32 || ((E->max_m + 7) / 8) * ((E->max_n + 7) / 8) >= 4) &&
               launch_big_f64(std::max(2, ((E->max_m + 7) / 8 + 1) / 2), std::max(2, ((E->max_n + 7) / 8 + 1) / 2), (unsigned)(8 * E->order_len), st,
                              E->descs.p, nblk, E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                              static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta,
                              skip_empty | (E->use_big == 2 ? 4 : 0), E->order.p)) {
      // blocks of 33 ... 80 (or an inner dimension above 32): one workgroup per C block, operand slabs shared through LDS (mm_numeric_f64_big.h)
      snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f64_big<%d,%d>", std::max(2, ((E->max_m + 7) / 8 + 1) / 2),
               std::max(2, ((E->max_n + 7) / 8 + 1) / 2));
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
      int grp_rc = 1;
#ifdef DBCSR_AMD_EXPERIMENTS
      if (E->use_hot && E->hot_m > 0 && E->hot_m == E->hot_n && E->hot_m == E->hot_k && E->f32_direct && E->f32_group != 0 && E->min_k == E->max_k &&
          E->max_k == E->hot_k && (E->f32_group > 0 || (E->nproducts >= 16 * nblk && nblk >= 1024))) {
        grp_rc = run_group_f32(E, E->f32_group > 0 ? E->f32_group : 4, reuse, st, a, b, c_in, c_out, (float)alpha, (float)beta, skip_empty);
        if (grp_rc < 0) return -1;
      }
#endif
      if (grp_rc == 0) {
        // the C blocks of other sizes (tail block row / column): the one-wave-per-block kernel, told to leave the dominant size alone
        if (E->hot_cnt_m < nbr || E->hot_cnt_n < b->nblkcols)
          launch_hot_f32_direct(E->hot_m, E->hot_n, E->hot_k, dim3(nwg_o), ww, st, E->descs.p, nblk, E->entries.p, static_cast<const float*>(a->data),
                                static_cast<const float*>(b->data), static_cast<float*>(c_out->data), static_cast<const float*>(c_in->data),
                                (float)alpha, (float)beta, skip_empty | 2, E->order.p);
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32_group<%d,%d,%d;%d>", E->hot_m, E->hot_n, E->hot_k, E->group_R);
      } else if (E->use_hot && E->hot_m > 0 && E->f32_direct &&
          launch_hot_f32_direct(E->hot_m, E->hot_n, E->hot_k, dim3(nwg_o), ww, st, E->descs.p, nblk, E->entries.p, static_cast<const float*>(a->data),
                                static_cast<const float*>(b->data), static_cast<float*>(c_out->data), static_cast<const float*>(c_in->data),
                                (float)alpha, (float)beta, skip_empty, E->order.p,
                                E->f32_direct >= 2 && E->hot_cnt_m == nbr && E->hot_cnt_n == b->nblkcols)) {
        snprintf(E->last_kernel, sizeof E->last_kernel, "mm_numeric_f32_direct<%d,%d,%d>", E->hot_m, E->hot_n, E->hot_k);
      } else if (E->use_hot && E->hot_m > 0 &&
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


#include "mm_engine_ops.h"   // init_c, crop, filter, checksum, fill, transpose, twin moves, statistics

}  // extern "C"
