// mm_engine_env.h -- part of mm_engine.hip (included inside namespace dbcsr_amd, after struct Engine): every environment switch of the engine, read ONCE
// per engine, when it is created (dbcsr_amd_mm_create) -- nothing on the multiply path calls getenv.  Shipping build: the switches that select among
// kernels that ship (A / B measurements, tests that force a kernel family); lab build: + the switches of the experimental dataflows.
#ifndef DBCSR_AMD_MM_ENGINE_ENV_H
#define DBCSR_AMD_MM_ENGINE_ENV_H

static void engine_read_env(Engine* E) {
  if (const char* k = getenv("DBCSR_AMD_MM_KERNEL")) {
    E->use_lds = strcmp(k, "direct") != 0;
    E->use_pipe = strcmp(k, "pipe") == 0 ? 1 : (strcmp(k, "lds1") == 0 ? 0 : -1);
#ifdef DBCSR_AMD_EXPERIMENTS
    if (strncmp(k, "dma", 3) == 0 && k[3] >= '2' && k[3] <= '4') E->dma_stages = k[3] - '0';
#endif
  }
  if (const char* k = getenv("DBCSR_AMD_MM_PIPE_G")) E->pipe_g = std::max(1, atoi(k));
  if (const char* k = getenv("DBCSR_AMD_MM_CLASSES")) E->use_classes = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_WORK")) E->use_work = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_WG_WAVES")) {
    const int w = atoi(k);
    if (w == 1 || w == 2 || w == 4) E->wg_waves = w;
  }
  if (const char* k = getenv("DBCSR_AMD_MM_PLAN")) E->use_plan = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_HOT")) E->use_hot = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TINY")) E->use_tiny = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_SMALL")) E->use_small = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_SMALL_G")) E->small_group = std::min(64, std::max(0, atoi(k)));
  if (const char* k = getenv("DBCSR_AMD_MM_F32_DIRECT")) E->f32_direct = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_BIG")) E->use_big = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_MID")) E->use_mid = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_F32_GROUP")) {
    const int r = atoi(k);
    E->f32_group = (r >= 2 && r <= 4) ? r : (r < 0 ? -1 : 0);
  }
  if (const char* k = getenv("DBCSR_AMD_MM_F64_GROUP")) E->f64_group = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_GROUP_PANEL_MB")) E->group_panel_bytes = (int64_t)atoll(k) << 20;
  if (const char* k = getenv("DBCSR_AMD_MM_SYMBOLIC")) {
    E->force_word_kernels = strcmp(k, "word") == 0;
    E->force_symbolic = strcmp(k, "word") == 0 ? 1 : (strcmp(k, "grid") == 0 ? 2 : (strcmp(k, "rows") == 0 ? 3 : 0));
  }
  if (const char* k = getenv("DBCSR_AMD_MM_PANEL_MB")) E->panel_bytes = (int64_t)atoll(k) << 20;
#ifdef DBCSR_AMD_EXPERIMENTS
  // ---- the lab build's switches (every one selects something that was measured and does not win; see the top of this file) ----
  if (const char* k = getenv("DBCSR_AMD_MM_CLASS_G")) {
    const int g = atoi(k);
    E->class_g = (g == 2 || g == 4 || g == 8) ? g : 1;
  }
  if (const char* k = getenv("DBCSR_AMD_MM_DBG")) E->dbg = atoi(k);
  {
    const char* k = getenv("DBCSR_AMD_MM_POISON");  // (process-wide: the engines created from now on)
    g_devbuf_poison = k ? (atoi(k) & 255) : -1;
  }
  if (const char* k = getenv("DBCSR_AMD_MM_HOT_VARIANT")) E->hot_variant = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_HOT_PERSISTENT")) E->hot_persistent = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_HOT_XCDS")) E->hot_xcd_mask = (unsigned)strtoul(k, nullptr, 0) & 0xffu;
  if (const char* k = getenv("DBCSR_AMD_MM_TILE")) E->use_tile = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_WINDOW")) E->tile_window = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_RDV")) E->tile_rdv = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_PUB")) E->tile_pub = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_PREFETCH")) E->tile_prefetch = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_KNOBS")) E->tile_knobs = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_TILE_SHAPE")) E->tile_shape = atoi(k) == 1 ? 1 : 0;
  if (const char* k = getenv("DBCSR_AMD_MM_BAND")) E->use_band = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_BAND_DEPTH")) E->band_depth = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_BAND_BPOL")) E->band_bpol = atoi(k) == 1 ? 1 : 0;
  if (const char* k = getenv("DBCSR_AMD_MM_BAND_KNOBS")) E->band_knobs = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_BAND_WINDOW")) E->band_window = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_BAND_SHAPE")) E->band_shape = atoi(k) == 0 ? 0 : 1;
  if (const char* k = getenv("DBCSR_AMD_MM_LDS_PAD")) E->lds_pad = atoi(k);
  if (const char* k = getenv("DBCSR_AMD_MM_CLASS_STREAMS")) E->class_streams = std::min(4, std::max(1, atoi(k)));
  if (const char* k = getenv("DBCSR_AMD_MM_ROW_GROUP")) E->row_group = atoi(k);
#endif
}

#endif
