// mm_engine_plan.h -- part of mm_engine.hip (included inside namespace dbcsr_amd): plan reuse -- does the incoming multiply have the index arrays of the
// last one (plan_matches: by address + stamp when trusted, else one comparison kernel), and saving them (plan_save).  DESIGN.md section 3.3.
#ifndef DBCSR_AMD_MM_ENGINE_PLAN_H
#define DBCSR_AMD_MM_ENGINE_PLAN_H

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
  if (E->plan_trusted && a->index_stamp && b->index_stamp && c_in->index_stamp && a->index_stamp == E->plan_stamps[0] &&
      b->index_stamp == E->plan_stamps[1] && c_in->index_stamp == E->plan_stamps[2]) {
    // same generation of the same arrays: the address test only guards against a caller that stamps carelessly
    bool same = true;
    for (int i = 0; i < 12; ++i) same = same && ptr[i] == E->plan_ptrs[i];
    if (same) return 1;
  }
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
  if (*E->plan_host_flag != 0) return 0;
  // equal arrays at (possibly) other addresses or of another generation: from now on THESE are the arrays the plan is known to fit, so a
  // caller that keeps them (a loop that passes its previous result back in) gets the cheap test next time
  for (int i = 0; i < 12; ++i) E->plan_ptrs[i] = ptr[i];
  E->plan_stamps[0] = a->index_stamp, E->plan_stamps[1] = b->index_stamp, E->plan_stamps[2] = c_in->index_stamp;
  return 1;
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
  for (int i = 0; i < 12; ++i) E->plan_ptrs[i] = ptr[i];
  E->plan_stamps[0] = a->index_stamp, E->plan_stamps[1] = b->index_stamp, E->plan_stamps[2] = c_in->index_stamp;
  E->plan_saved = true;
  return 0;
}

#endif
