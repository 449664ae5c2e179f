// mm_engine_lab.h -- part of mm_engine.hip, LAB build only (included inside namespace dbcsr_amd): host side of the experimental dataflows (group, tile,
// band kernels): built, parity-green, measured slower; DESIGN.md section 6.
#ifndef DBCSR_AMD_MM_ENGINE_LAB_H
#define DBCSR_AMD_MM_ENGINE_LAB_H

#ifdef DBCSR_AMD_EXPERIMENTS
// fp32 group kernel: 0 = launched, 1 = does not apply here (the caller runs the one-wave-per-block kernel), < 0 = error
static int run_group_f32(Engine* E, int R, bool reuse, hipStream_t st, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
                         dbcsr_amd_bcsr* c_out, float alpha, float beta, int skip_empty) {
  const int S = E->hot_m, nbr = E->nbr, nbc = b->nblkcols;
  if (!(S == 16 || S == 24 || S == 32) || R < 2 || R > 4 || nbr <= 0 || nbc <= 0) return 1;
  const int ng = (nbr + R - 1) / R, ngx = (ng + 7) / 8;
  if ((int64_t)ngx * nbc >= (1ll << 30)) return 1;
  if (!(reuse && E->group_built && E->group_R == R)) {
    if (E->group_flag.ensure(4) || E->groups.ensure((size_t)ng * nbc * R + 1)) return -1;
    ACC_CHECK(hipMemsetAsync(E->group_flag.p, 0, sizeof(int), st));
    group_check_ascending(st, static_cast<const int64_t*>(b->blk_p), (int64_t)b->nblks, E->group_flag.p);
    int* hflag = reinterpret_cast<int*>(E->host_scalars + 12);
    ACC_CHECK(hipMemcpyAsync(hflag, E->group_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    ACC_CHECK(hipStreamSynchronize(st));
    E->b_monotone = *hflag == 0;
    E->group_R = R;
    E->group_built = true;
    if (E->b_monotone) group_build_table(st, c_out->row_p, c_out->col_i, E->descs.p, nbr, nbc, R, S, E->groups.p);
  }
  if (!E->b_monotone) return 1;
  GroupGeom G;
  G.nbc = nbc, G.ng = ng, G.ngx = ngx;
  const int64_t b_bytes = (int64_t)b->nblks * S * S * (int64_t)sizeof(float);
  int np = (int)std::min<int64_t>(std::max<int64_t>(1, (b_bytes + E->panel_bytes - 1) / E->panel_bytes), (int64_t)nbc);
  G.pw = (nbc + np - 1) / np;
  G.np = (nbc + G.pw - 1) / G.pw;
  const unsigned nwg = 8u * (unsigned)(((int64_t)ngx * nbc + 3) / 4);
  const float* ad = static_cast<const float*>(a->data);
  const float* bd = static_cast<const float*>(b->data);
  float* cd = static_cast<float*>(c_out->data);
  const float* cid = static_cast<const float*>(c_in->data);
  return group_f32_launch(S, R, nwg, st, E->descs.p, E->entries.p, ad, bd, cd, cid, alpha, beta, skip_empty, E->groups.p, G);
}

// fp64 group kernel (mm_group64.h): 0 = launched, 1 = does not apply here (the caller runs the one-wave-per-block kernel), < 0 = error
static int run_group_f64(Engine* E, int R, bool reuse, hipStream_t st, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
                         dbcsr_amd_bcsr* c_out, double alpha, double beta, int skip_empty) {
  const int S = E->hot_m, nbr = E->nbr, nbc = b->nblkcols;
  if (!group64_has_kernel(S, R) || nbr <= 0 || nbc <= 0) return 1;
  const int ng = (nbr + R - 1) / R, ngx = (ng + 7) / 8;
  const int64_t ngj = (int64_t)ng * nbc;
  if ((int64_t)ngx * nbc >= (1ll << 28)) return 1;
  if (!(reuse && E->group_built && E->group_R == R)) {
    if (E->group_flag.ensure(4) || E->groups.ensure((size_t)ngj * R + 1) || E->group_cnt.ensure((size_t)ngj + 1) || E->group_work.ensure((size_t)ngj + 1) ||
        E->group_start.ensure((size_t)ngj + 2))
      return -1;
    ACC_CHECK(hipMemsetAsync(E->group_flag.p, 0, sizeof(int), st));
    group_check_ascending(st, static_cast<const int64_t*>(b->blk_p), (int64_t)b->nblks, E->group_flag.p);
    int* hflag = reinterpret_cast<int*>(E->host_scalars + 12);
    ACC_CHECK(hipMemcpyAsync(hflag, E->group_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    ACC_CHECK(hipStreamSynchronize(st));
    E->b_monotone = *hflag == 0;
    E->group_R = R;
    E->group_built = true;
    if (E->b_monotone) {
      // the merged lists hold at most as many records as there are products
      if (E->group_entries.ensure((size_t)E->nproducts + 4)) return -1;
      group_build_table(st, c_out->row_p, c_out->col_i, E->descs.p, nbr, nbc, R, S, E->groups.p);
      group64_count(st, E->groups.p, E->descs.p, ngj, R, E->group_cnt.p);
      if (exclusive_scan<int64_t>(E, E->group_cnt.p, ngj, E->group_start.p, nullptr, true, st)) return -1;
      group64_merge(st, E->groups.p, E->descs.p, E->entries.p, ngj, R, S, E->group_start.p, E->group_work.p, E->group_entries.p);
    }
  }
  if (!E->b_monotone) return 1;
  GroupGeom G;
  G.nbc = nbc, G.ng = ng, G.ngx = ngx;
  const int64_t b_bytes = (int64_t)b->nblks * S * S * (int64_t)sizeof(double);
  const int64_t pb = E->group_panel_bytes > 0 ? E->group_panel_bytes : E->panel_bytes;
  int np = (int)std::min<int64_t>(std::max<int64_t>(1, (b_bytes + pb - 1) / pb), (int64_t)nbc);
  G.pw = (nbc + np - 1) / np;
  G.np = (nbc + G.pw - 1) / G.pw;
  const int has_tail = (E->min_k != E->max_k || E->max_k != S) ? 1 : 0;
  return group64_launch(S, R, st, E->descs.p, E->entries.p, static_cast<const double*>(a->data), static_cast<const double*>(b->data),
                        static_cast<double*>(c_out->data), static_cast<const double*>(c_in->data), alpha, beta, skip_empty, has_tail, E->group_work.p,
                        E->group_entries.p, G);
}
#endif

#ifdef DBCSR_AMD_EXPERIMENTS
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
  if (!tile_shape(E->tile_shape, &G.tr, &G.tc, &G.wg_waves)) return 1;
  G.nTR = (G.nfr + G.tr - 1) / G.tr;
  G.nTC = (G.nfc + G.tc - 1) / G.tc;
  G.team_rows = std::max(1, cu_per_xcd * G.wg_waves / kTeamCols);
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
  P.knobs = E->tile_knobs;
  P.times = nullptr;
  if (E->tile_knobs & 32) {
    if (E->tile_times.ensure(8)) return -1;
    ACC_CHECK(hipMemsetAsync(E->tile_times.p, 0, 8 * sizeof(unsigned long long), st));
    P.times = E->tile_times.p;
  }
  ACC_CHECK(hipEventRecord(E->ev[1], st));  // the timed numeric launch starts here (the index work above counts as fill time)
  if (tile_launch(S_, S_, S_, E->tile_rdv, E->tile_shape, (unsigned)(8 * cu_per_xcd), st, P)) return -1;
  if (tile_launch_remainder(S_, S_, st, G, E->tdescs.p, E->tentries.p, P.a_data, P.b_data, P.c_out, alpha)) return -1;
  return check(hipGetLastError(), "run_tile_f64", __FILE__, __LINE__);
}

// The band dataflow (mm_band.h) for the C blocks of the dominant size: bitmaps of A and of B transposed, sub-tile descriptors, the
// product lists in sweep order (count, scan, fill), the persistent kernel, the products with inner blocks of another size.  The
// caller then runs the exact-size kernel over the C blocks of the other sizes.  descs[] and C_out's index are already filled.
template <int S_>
static int run_band_f64(Engine* E, hipStream_t st, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
                        dbcsr_amd_bcsr* c_out, double alpha, double beta) {
  const int nbr = a->nblkrows, nbk = a->nblkcols, nbc = b->nblkcols, W = E->W, Wk = (nbk + 31) / 32;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    ACC_CHECK(hipGetDevice(&dev));
    ACC_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  if (band_lds_bytes(S_, S_, S_, E->band_shape, E->band_depth) == 0) return 1;
  BandGeom G;
  G.nfr = E->hot_cnt_m;
  G.nfc = E->hot_cnt_n;
  if (G.nfr <= 0 || G.nfc <= 0 || !band_shape(E->band_shape, &G.waves, &G.tr, &G.tc)) return 1;
  G.nBR = (G.nfr + G.waves * G.tr - 1) / (G.waves * G.tr);
  G.nBC = (G.nfc + G.tc - 1) / G.tc;
  if ((int64_t)G.nBR * G.nBC > 0x3fffffff) return 1;
  G.ntiles = G.nBR * G.nBC;
  G.cu_per_xcd = std::min(32, std::max(1, n_cu / 8));
  G.max_i = 1;
  for (int x = 0; x < 8; ++x) G.max_i = std::max(G.max_i, (int)((G.lo(x + 1) - G.lo(x) + G.cu_per_xcd - 1) / G.cu_per_xcd));
  G.kshift = 0;
  while ((nbk >> G.kshift) >= 4096) ++G.kshift;
  G.kspan = (nbk >> G.kshift) + 1;
  if ((int64_t)(G.max_i + 1) * G.kspan >= 0x7ff00000ll) return 1;  // sweep positions would overflow: not a band case
  const int nwg = 8 * G.cu_per_xcd;
  const int64_t nsub = (int64_t)G.waves * G.ntiles, npl = (int64_t)nwg * G.waves * G.max_i, nps = (int64_t)nwg * G.max_i;
  const bool reuse = E->plan_hit && E->plan_numeric && E->band_built && E->band_geom.waves == G.waves && E->band_geom.ntiles == G.ntiles;
  if (E->band_flags.ensure(4) || E->band_prog.ensure(8 * 512)) return -1;
  ACC_CHECK(hipMemsetAsync(E->band_flags.p, 0, sizeof(int) * 4, st));
  ACC_CHECK(hipMemsetAsync(E->band_prog.p, 0, sizeof(unsigned) * 8 * 512, st));
  if (!reuse) {
    E->band_built = false;
    if (E->a_bm.ensure((size_t)nbr * Wk + 1) || E->a_pre.ensure((size_t)nbr * Wk + 1) || E->bt_bm.ensure((size_t)nbc * Wk + 1) ||
        E->tile_rows.ensure((size_t)nbr + 1) || E->tile_cols.ensure((size_t)nbc + 1) || E->band_descs_buf.ensure((size_t)nsub + 1) ||
        E->band_sub_cnt.ensure((size_t)nsub + 1) || E->band_cnt_rem.ensure((size_t)nsub + 1) || E->band_rem_start.ensure((size_t)nsub + 2) ||
        E->band_cnt_list.ensure((size_t)npl + 1) || E->band_list_off.ensure((size_t)npl + 2) || E->band_cnt_b.ensure((size_t)nps + 1) ||
        E->band_seq_off.ensure((size_t)nps + 2))
      return -1;
    ACC_CHECK(hipMemsetAsync(E->a_bm.p, 0, sizeof(uint32_t) * (size_t)nbr * Wk, st));
    ACC_CHECK(hipMemsetAsync(E->bt_bm.p, 0, sizeof(uint32_t) * (size_t)nbc * Wk, st));
    ACC_CHECK(hipMemsetAsync(E->band_cnt_list.p, 0, sizeof(int) * (size_t)npl, st));
    ACC_CHECK(hipMemsetAsync(E->band_cnt_b.p, 0, sizeof(int) * (size_t)nps, st));
    hipLaunchKernelGGL(bitmap_from_index, grid_for((int64_t)nbr * 64), dim3(256), 0, st, a->row_p, a->col_i, nbr, Wk, E->a_bm.p);
    hipLaunchKernelGGL(row_prefix, grid_for((int64_t)nbr * 64), dim3(256), 0, st, E->a_bm.p, nbr, Wk, E->a_pre.p, (int*)nullptr);
    hipLaunchKernelGGL(tile_bitmap_transposed, grid_for((int64_t)nbk * 64), dim3(256), 0, st, b->row_p, b->col_i, nbk, Wk, E->bt_bm.p);
    hipLaunchKernelGGL(tile_select, dim3(1), dim3(64), 0, st, a->row_blk_size, nbr, S_, E->tile_rows.p, nbr);
    hipLaunchKernelGGL(tile_select, dim3(1), dim3(64), 0, st, b->col_blk_size, nbc, S_, E->tile_cols.p, nbc);
    hipLaunchKernelGGL(band_descs, grid_for(nsub * 16), dim3(256), 0, st, G, E->tile_rows.p, E->tile_cols.p, E->c_bm.p, E->c_pre.p, c_out->row_p, W,
                       E->descs.p, E->band_descs_buf.p, E->band_sub_cnt.p);
    hipLaunchKernelGGL((band_lists<false>), grid_for(nsub * 64), dim3(256), 0, st, G, E->tile_rows.p, E->tile_cols.p, nbk, Wk, E->a_bm.p, E->a_pre.p,
                       a->row_p, a->blk_p, E->bt_bm.p, W, E->b_bm.p, E->b_pre.p, b->row_p, b->blk_p, a->col_blk_size, S_, E->band_cnt_list.p,
                       E->band_cnt_b.p, E->band_cnt_rem.p, (const int64_t*)nullptr, (const int64_t*)nullptr, (const int64_t*)nullptr,
                       E->band_sub_cnt.p, (BandEntry*)nullptr, (BandRem*)nullptr, E->band_flags.p + 1);
    if (exclusive_scan<int64_t>(E, E->band_cnt_list.p, npl, E->band_list_off.p, nullptr, true, st)) return -1;
    if (exclusive_scan<int64_t>(E, E->band_cnt_b.p, nps, E->band_seq_off.p, nullptr, true, st)) return -1;
    if (exclusive_scan<int64_t>(E, E->band_cnt_rem.p, nsub, E->band_rem_start.p, nullptr, true, st)) return -1;
    hipLaunchKernelGGL(band_max_seq, grid_for(nwg), dim3(256), 0, st, G, nwg, E->band_seq_off.p, E->band_flags.p + 2);
    // list sizes to the host (once per plan: a multiply that reuses the plan comes nowhere near this)
    ACC_CHECK(hipMemcpyAsync(E->host_scalars + 8, E->band_list_off.p + npl, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    ACC_CHECK(hipMemcpyAsync(E->host_scalars + 9, E->band_rem_start.p + nsub, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    ACC_CHECK(hipMemcpyAsync(E->host_scalars + 10, E->band_flags.p + 2, sizeof(int), hipMemcpyDeviceToHost, st));
    ACC_CHECK(hipStreamSynchronize(st));
    E->band_nlist = E->host_scalars[8];
    E->band_nrem = E->host_scalars[9];
    const int max_seq = *reinterpret_cast<const int*>(E->host_scalars + 10);
    if (max_seq >= (1 << 23) - 64) return 1;  // the entries carry 23 bits of the sequence number: not a band case
    if (E->band_entries.ensure((size_t)E->band_nlist + 1) || E->band_rem.ensure((size_t)E->band_nrem + 1)) return -1;
    hipLaunchKernelGGL((band_lists<true>), grid_for(nsub * 64), dim3(256), 0, st, G, E->tile_rows.p, E->tile_cols.p, nbk, Wk, E->a_bm.p, E->a_pre.p,
                       a->row_p, a->blk_p, E->bt_bm.p, W, E->b_bm.p, E->b_pre.p, b->row_p, b->blk_p, a->col_blk_size, S_, (int*)nullptr,
                       (int*)nullptr, (int*)nullptr, E->band_list_off.p, E->band_seq_off.p, E->band_rem_start.p, E->band_sub_cnt.p,
                       E->band_entries.p, E->band_rem.p, E->band_flags.p + 1);
    E->band_geom = G;
    E->band_built = true;
  }
  BandArgs P;
  P.descs = E->band_descs_buf.p;
  P.entries = E->band_entries.p;
  P.list_off = E->band_list_off.p;
  P.a_data = static_cast<const double*>(a->data);
  P.b_data = static_cast<const double*>(b->data);
  P.c_out = static_cast<double*>(c_out->data);
  P.c_in = static_cast<const double*>(c_in->data);
  P.alpha = alpha;
  P.beta = beta;
  P.G = G;
  P.flags = E->band_flags.p;
  P.prog = E->band_prog.p;
  P.window = E->band_window > 0 ? std::max(1, E->band_window >> G.kshift) : 0;
  P.knobs = E->band_knobs;
  P.times = nullptr;
  if (E->band_knobs & 1) {
    if (E->band_times.ensure(16)) return -1;
    ACC_CHECK(hipMemsetAsync(E->band_times.p, 0, 16 * sizeof(unsigned long long), st));
    P.times = E->band_times.p;
  }
  ACC_CHECK(hipEventRecord(E->ev[1], st));  // the timed numeric launch starts here (the index work above counts as fill time)
  if (band_launch(S_, S_, S_, E->band_shape, E->band_depth, E->band_bpol, (unsigned)nwg, st, P)) return -1;
  if (E->band_nrem > 0 &&
      band_launch_remainder(S_, S_, st, nsub, E->band_descs_buf.p, E->band_rem_start.p, E->band_rem.p, P.a_data, P.b_data, P.c_out, alpha))
    return -1;
  return check(hipGetLastError(), "run_band_f64", __FILE__, __LINE__);
}
#endif  // DBCSR_AMD_EXPERIMENTS

#endif
