// mm_engine_ops.h -- part of mm_engine.hip (included inside extern "C", after the multiply's own entry points): the C-ABI operations around the multiply --
// dbcsr_amd_mm_init_c, crop / window scale (submatrix limits), block filter, checksum, synthetic fill, transpose, desymmetrize / twin moves, statistics,
// kernel names, plan switches.
#ifndef DBCSR_AMD_MM_ENGINE_OPS_H
#define DBCSR_AMD_MM_ENGINE_OPS_H

int dbcsr_amd_mm_init_c(void* handle, libsmm_acc_data_t datatype, double beta, const dbcsr_amd_bcsr* c_in, dbcsr_amd_bcsr* c_out,
                        void* stream) {
  Engine* E = static_cast<Engine*>(handle);
  if (E) E->plan_numeric = false;  // the descriptors are rewritten below: a numeric phase that follows fills its lists again (the plan itself stands)
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
  if (have_norms && E->unwritten_below > eps * eps) {
    fprintf(stderr, "dbcsr_amd_bcsr_filter_count: this matrix was multiplied with a final filter of eps^2 = %g announced (dbcsr_amd_mm_expect_filter); "
                    "blocks below that were not written and cannot be kept with eps^2 = %g\n", E->unwritten_below, eps * eps);
    return -3;
  }
  E->norms_data = nullptr;
  E->unwritten_below = 0.0;
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

int dbcsr_amd_mm_expect_filter(void* handle, double eps) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E || !(eps >= 0.0)) return -1;
  E->drop_pending = eps * eps;
  return 0;
}

int dbcsr_amd_mm_trust_plan(void* handle, int on) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E) return -1;
  E->plan_trusted = on != 0;
  return 0;
}

int dbcsr_amd_mm_plan_stats(void* handle, int64_t* reused, int64_t* built) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E) return -1;
  if (reused) *reused = E->plan_hits;
  if (built) *built = E->plan_misses;
  return 0;
}

#ifdef DBCSR_AMD_EXPERIMENTS
int dbcsr_amd_mm_tile_stats(void* handle, int* waves_gave_up, int* list_mismatches) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E) return -1;
  if (strncmp(E->last_kernel, "mm_numeric_f64_tile", 19) != 0 || !E->tile_flags.p) return 1;
  int h[4] = {0, 0, 0, 0};
  ACC_CHECK(hipDeviceSynchronize());
  ACC_CHECK(hipMemcpy(h, E->tile_flags.p, sizeof h, hipMemcpyDeviceToHost));
  if (waves_gave_up) *waves_gave_up = h[0];
  if (list_mismatches) *list_mismatches = h[1];
  if ((E->tile_knobs & 32) && E->tile_times.p) {
    unsigned long long t[8];
    ACC_CHECK(hipMemcpy(t, E->tile_times.p, sizeof t, hipMemcpyDeviceToHost));
    const double w = t[5] ? (double)t[5] : 1.0;
    fprintf(stderr, "dbcsr_amd tile kernel, mean per wave [ms]: total %.3f = window waits %.3f + operand waits %.3f + multiplies %.3f + epilogues %.3f + rest %.3f (%llu waves)\n",
            t[0] / w * 1e-5, t[1] / w * 1e-5, t[2] / w * 1e-5, t[3] / w * 1e-5, t[4] / w * 1e-5, ((double)t[0] - t[1] - t[2] - t[3] - t[4]) / w * 1e-5, t[5]);
  }
  if (getenv("DBCSR_AMD_MM_TILE_VERBOSE"))
    fprintf(stderr, "dbcsr_amd tile kernel: %d waves gave up, %lld reads of the team counters, %lld products waited for the window (of %lld)\n", h[0],
            16ll * h[2], 16ll * h[3], (long long)E->nproducts);
  return 0;
}

int dbcsr_amd_mm_band_stats(void* handle, int* waits_gave_up, int* list_mismatches) {
  Engine* E = static_cast<Engine*>(handle);
  if (!E) return -1;
  if (strncmp(E->last_kernel, "mm_numeric_f64_band", 19) != 0 || !E->band_flags.p) return 1;
  int h[4] = {0, 0, 0, 0};
  ACC_CHECK(hipDeviceSynchronize());
  ACC_CHECK(hipMemcpy(h, E->band_flags.p, sizeof h, hipMemcpyDeviceToHost));
  if (waits_gave_up) *waits_gave_up = h[0];
  if (list_mismatches) *list_mismatches = h[1];
  if ((E->band_knobs & 1) && E->band_times.p) {
    unsigned long long t[16];
    ACC_CHECK(hipMemcpy(t, E->band_times.p, sizeof t, hipMemcpyDeviceToHost));
    const double w = t[5] ? (double)t[5] : 1.0;
    fprintf(stderr,
            "dbcsr_amd band kernel, mean per wave [ms]: total %.3f = window waits %.3f + issue and waits for A %.3f + waits for B %.3f + multiplies %.3f + "
            "epilogues %.3f + rest %.3f (%llu waves; %llu of %lld products waited for their B block, %llu fetched it themselves at the last moment; %llu "
            "fetches waited for the window, %llu reads of the team's counters, %d waves switched the throttle off; %lld list entries, shape %d, ring of %d, window %d)\n",
            t[0] / w * 1e-5, t[8] / w * 1e-5, t[1] / w * 1e-5, t[2] / w * 1e-5, t[3] / w * 1e-5, t[4] / w * 1e-5,
            ((double)t[0] - t[1] - t[2] - t[3] - t[4] - t[8]) / w * 1e-5, t[5], t[7], (long long)E->nproducts, t[6], t[9], t[10], h[3], (long long)E->band_nlist,
            E->band_shape, E->band_depth, E->band_window);
  }
  return 0;
}
#endif  // DBCSR_AMD_EXPERIMENTS

#endif
