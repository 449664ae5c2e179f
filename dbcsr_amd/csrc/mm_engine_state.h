// mm_engine_state.h -- part of mm_engine.hip (included inside namespace dbcsr_amd): the plan-comparison kernel, struct Engine (buffers, switches, the saved
// plan of the last symbolic phase) and the small host helpers everything else uses.
#ifndef DBCSR_AMD_MM_ENGINE_STATE_H
#define DBCSR_AMD_MM_ENGINE_STATE_H

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
  // fp32: a wave owns R C blocks of one block column and shares B among them (mm_group.h).  DBCSR_AMD_MM_F32_GROUP = 2 / 3 / 4: that R
  // whenever the kernel applies; -1: R = 4 when C blocks have at least 16 products on average; 0 / unset: off -- measured (gpurun_out/r05_s04:
  // 32768^2 at 20 % fill 32.1 ms against 28.6 for one wave per block, config 5 2125 against 1836 ms) it trades B blocks over the fabric for
  // A rows that no longer fit the XCD's L2 and for occupancy (3 waves per SIMD instead of 5), and loses
  int f32_group = 0, group_R = 0;
  bool group_built = false, b_monotone = false;
  DevBuf<int> groups, group_flag;
  // fp64 (round 6, mm_group64.h): DBCSR_AMD_MM_F64_GROUP = 2 ... 6: a wave owns that many C blocks of one block column whenever the kernel
  // applies; 0 / unset: off.  DBCSR_AMD_MM_GROUP_PANEL_MB: target size of a B column panel of the group launch (0: panel_bytes)
  int f64_group = 0;
  int64_t group_panel_bytes = 0;
  DevBuf<int> group_cnt;
#ifdef DBCSR_AMD_EXPERIMENTS
  DevBuf<GWork> group_work;
  DevBuf<GEntry> group_entries;
#endif
  DevBuf<int64_t> group_start;
  int use_mid = 1;     // DBCSR_AMD_MM_MID=0: blocks of 33 ... 40 through the workgroup kernel mm_numeric_f64_big instead of the one-wave kernel mm_numeric_f64_mid
  int use_big = 1;     // DBCSR_AMD_MM_BIG=0: blocks above 32 through the one-wave-per-block kernel of rounds 1-4 (mm_numeric_f64) instead of mm_numeric_f64_big
  int f32_direct = 1;  // (2: + the slim-LDS launch when every C block has the dominant size -- more waves per CU, measured 0-4 % slower: the
                       // kernel is fabric-bound, gpurun_out/r05_s17 --, 1: never slim) DBCSR_AMD_MM_F32_DIRECT=0: the fp32 exact-size kernel that stages both operands in LDS (rounds 1-4) instead of the direct form
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
  double drop_pending = 0.0;       // dbcsr_amd_mm_expect_filter: eps^2 of the final block filter announced for the next numeric phase (0: none)
  double unwritten_below = 0.0;    // the C of the last numeric phase lacks the blocks with ||blk||^2 below this (their norms are in norms64)
  int units_m = 0, units_cnt_m = 0, units_n = 0, units_cnt_n = 0;   // most frequent size of C's rows / columns in units of 4 (sizes up to 48) and how often (block_size_stats)
  int use_tiny = 1;                     // DBCSR_AMD_MM_TINY=0: no packed kernel for blocks of at most 4 x 4
  int small_group = 0;                  // DBCSR_AMD_MM_SMALL_G: C blocks a wave of the small-block kernel takes one after the other (0: eight when the lists are short, else one)
  int use_small = 2;                    // DBCSR_AMD_MM_SMALL=0: no one-tile kernel for multiplies whose block dimensions are all <= 8 (mm_numeric_f64_small.h); 2 (default) / 3 / 4 / 6 / 8: products in flight per wave
  int use_hot = 1;                      // DBCSR_AMD_MM_HOT=0: never use the exact-size kernels
  int lds_pad = 0;                      // DBCSR_AMD_MM_LDS_PAD: extra LDS bytes per workgroup (occupancy experiments)
  int64_t panel_bytes = 256ll << 20;  // DBCSR_AMD_MM_PANEL_MB: target size of a B column panel (config 2, round 3: 160 / 200 / 256 / 320 / 400 MB ->
                                      // 18.97 / 18.76 / 18.63 / 18.95 / 19.04 ms, profiles/r03_panel_wgwaves_sweep.txt)
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
  int tile_shape = 0;  // DBCSR_AMD_MM_TILE_SHAPE: 0 = 3 x 3 C blocks per wave, two waves per SIMD; 1 = 4 x 3, one wave per SIMD, four-slot ring (mm_tile.h)
  int use_tile = 0, tile_window = 256, tile_rdv = 0, tile_pub = 1, tile_prefetch = 0, tile_knobs = 0;  // DBCSR_AMD_MM_TILE_PUB: progress stores written through (0) / left in L2 (1)
  int hot_cnt_m = 0, hot_cnt_k = 0, hot_cnt_n = 0;  // block rows / inner blocks / block columns of the dominant size
  DevBuf<uint32_t> a_bm, bt_bm, tile_prog;
  DevBuf<unsigned long long> tile_times;
  DevBuf<int> a_pre, tile_rows, tile_cols, tile_cnt, tile_flags;
  DevBuf<int64_t> tile_start;
#ifdef DBCSR_AMD_EXPERIMENTS
  DevBuf<TileDesc> tdescs;
  DevBuf<TileEntry> tentries;
#endif
  // plan reuse (plan_compare): device copies of the index arrays the last symbolic phase saw, C's index as the numeric phase emitted it
  int use_plan = 1;  // DBCSR_AMD_MM_PLAN=0: every multiply runs its symbolic phase
  bool plan_saved = false, plan_hit = false, plan_numeric = false;
  int plan_dims[3] = {0, 0, 0}, plan_retain = 0, plan_canonical = 0, plan_datatype = 0;
  int64_t plan_nblks[3] = {0, 0, 0};
  // dbcsr_amd_mm_trust_plan: index arrays at the ADDRESSES the saved plan saw are taken as unchanged (no comparison on the device, no
  // synchronisation): for callers that own their operands' index and never write it in place
  bool plan_trusted = false;
  const void* plan_ptrs[12] = {nullptr};
  uint64_t plan_stamps[3] = {0, 0, 0};  // index_stamp of A, B, C_in when the plan was saved (0: unknown generation, never trusted)
  DevBuf<int32_t> plan_words, plan_c_col_i;
  DevBuf<int64_t> plan_c_blk_p;
  DevBuf<int> plan_flag;
  int* plan_host_flag = nullptr;  // pinned
  dbcsr_amd_mm_counts plan_counts = {0, 0, 0, 0};
  bool work_built = false, tile_built = false, band_built = false;
#ifdef DBCSR_AMD_EXPERIMENTS
  TileGeom tile_geom = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // CU-wide C tiles, B shared in an LDS ring (mm_band.h): DBCSR_AMD_MM_BAND = 0 never, 1 automatic, 2 whenever the sizes allow;
  // DBCSR_AMD_MM_BAND_DEPTH = slots of the ring (12 | 16 | 20 | 22); DBCSR_AMD_MM_BAND_BPOL = 1: B copies with the nt hint;
  // DBCSR_AMD_MM_BAND_KNOBS bit 0: where the waves' time goes (printed by dbcsr_amd_mm_band_stats)
  // DBCSR_AMD_MM_BAND_WINDOW = k window of an XCD's waves (inner blocks; 0: no throttle)
  // DBCSR_AMD_MM_BAND_SHAPE: 0 = 8 waves x (3 x 3 C blocks), 1 = 16 waves x (2 x 2)
  int use_band = 0, band_shape = 1, band_depth = 20, band_bpol = 0, band_knobs = 0, band_window = 384;
  int64_t band_nlist = 0, band_nrem = 0;
  DevBuf<unsigned> band_prog;
#ifdef DBCSR_AMD_EXPERIMENTS
  BandGeom band_geom = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  DevBuf<BandDesc> band_descs_buf;
  DevBuf<BandEntry> band_entries;
  DevBuf<BandRem> band_rem;
#endif
  DevBuf<int> band_cnt_list, band_cnt_b, band_cnt_rem, band_sub_cnt, band_flags;
  DevBuf<int64_t> band_list_off, band_seq_off, band_rem_start;
  DevBuf<unsigned long long> band_times;
  long long plan_hits = 0, plan_misses = 0;
  int hot_persistent = 0;  // DBCSR_AMD_MM_HOT_PERSISTENT=1: the 23^3 kernel as persistent waves with a work counter per XCD (mm_numeric_f64.h)
  unsigned hot_xcd_mask = 0xffu;  // DBCSR_AMD_MM_HOT_XCDS: XCDs the persistent form runs on (experiments: the others' C blocks are NOT computed)
  DevBuf<unsigned> hot_counters;
  int hot_variant = 0;  // DBCSR_AMD_MM_HOT_VARIANT: 2 = exact-size kernel with unpaired ds_read_b64 fragment reads (23^3 only)
  int use_pipe = -1, pipe_g = 8;  // multi-block pipelined kernel: -1 automatic (short product lists only), DBCSR_AMD_MM_KERNEL=pipe|lds1 forces; DBCSR_AMD_MM_PIPE_G = blocks per wave
  // (m, n) classes (mixed block sizes, see order_count_cls): DBCSR_AMD_MM_CLASSES = 0 never, 1 automatic, 2 always when the sizes allow
  int use_classes = 1;
  int class_g = 1;  // DBCSR_AMD_MM_CLASS_G: C blocks per wave in the class kernels (1, 2, 4, 8)
  bool cls_mode = false;
  // DBCSR_AMD_MM_CLASS_STREAMS: the class launches of one multiply touch disjoint C blocks; with n > 1 they are spread over n streams
  // (the caller's + n - 1 of the engine's, forked / joined with events) so that the tail of one launch overlaps the body of the next
  int class_streams = 1;
  hipStream_t side_stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t fork_ev = nullptr, join_ev[3] = {nullptr, nullptr, nullptr};
  int cls_m[3] = {0, 0, 0}, cls_n[3] = {0, 0, 0}, cls_k[3] = {0, 0, 0};
  int64_t cls_len[kNumClasses] = {0}, cls_off[kNumClasses] = {0};
  DevBuf<int> cls_hist;
  DevBuf<unsigned char> cls_row, cls_col;
  DevBuf<int> cls_vpos, cls_vrow;   // class mode: the rows numbered again class after class (class_row_deal, mm_symbolic.h) and the inverse
  DevBuf<uint32_t> cls_col_bm;
  DevBuf<int64_t> cls_lens;
  int* cls_host_hist = nullptr;       // pinned: 3 x 33 size histograms
  int64_t* cls_host_lens = nullptr;   // pinned: 10 lengths, 10 offsets, total
  char last_kernel[96] = "";  // name of the numeric kernel of the last dbcsr_amd_mm_numeric (dbcsr_amd_mm_last_kernel)
  int dma_stages = 0;  // DBCSR_AMD_MM_KERNEL=dma2|dma3|dma4: LDS-DMA exact-size kernel with that many ring slots (0: off)
  int use_lds = 1;  // DBCSR_AMD_MM_KERNEL=direct selects the v1 kernel (A/B experiments)
  KPassMemo kpass_memo;  // dbcsr_amd_multiply's k-pass decision for the last stamped A operand (mm_api.hip)
};

KPassMemo* engine_kpass_memo(void* handle) { return handle ? &static_cast<Engine*>(handle)->kpass_memo : nullptr; }

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

#endif
