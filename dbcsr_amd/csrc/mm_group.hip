// mm_group.hip -- the kernels of the fp32 group dataflow (mm_group.h).  A translation unit of its own because it is compiled with
// -mllvm -structurizecfg-skip-uniform-regions (as mm_tile.hip): a step multiplies into one of R accumulator sets under wave-uniform
// branches, and without the option the compiler restructures those branches into flow blocks whose joins keep the accumulators in
// vector registers -- 16 v_accvgpr_write before and 16 v_accvgpr_read after EVERY product, 150-200 registers.
#include "mm_group.h"
#include "smm_core.h"

namespace dbcsr_amd {

template <int M, int N, int K, int R>
__global__ void __launch_bounds__(256) mm_numeric_f32_group(const Desc* __restrict__ descs, const Entry* __restrict__ entries,
                                                            const float* __restrict__ a_data, const float* __restrict__ b_data,
                                                            float* __restrict__ c_out, const float* __restrict__ c_in, float alpha, float beta,
                                                            int skip_empty, const int* __restrict__ groups, GroupGeom G) {
  static_assert(K % 8 == 0 && K >= 8 && K <= 32 && M <= 32 && N <= 32 && R >= 2 && R <= 4, "mm_numeric_f32_group: shape");
  extern __shared__ __attribute__((aligned(16))) char smem_raw_[];
  constexpr int PB = f32d_pitch(K), KH = K / 2, Q = KH / 4;
  constexpr int CB = (K * N * 4 + 1023) / 1024;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = (int)(blockDim.x >> 6);
  float* lds_b = reinterpret_cast<float*>(smem_raw_) + (size_t)wid * f32d_wave_floats(K);
  // position of this wave in its XCD's stream -> (panel, row group, column)
  const int xcd = (int)(blockIdx.x & 7u);
  const int s = (int)(blockIdx.x >> 3) * nw + wid;   // (the host keeps ngx * nbc below 2^31)
  const int full = G.ngx * G.pw;
  int p = s / full;
  if (p > G.np - 1) p = G.np - 1;
  const int rem = s - p * full;
  const int pwl = p == G.np - 1 ? G.nbc - p * G.pw : G.pw;
  const int gl = rem / pwl, j = p * G.pw + rem % pwl;
  const int g = xcd + 8 * gl;
  if (gl >= G.ngx || g >= G.ng) return;
  // the group's blocks and their lists.  Everything about the lists is wave-uniform and is pinned to scalar registers explicitly
  // (readfirstlane): left to itself the compiler kept this state in vector registers -- 202 of them.
  const int* gr = groups + ((int64_t)g * G.nbc + j) * R;
  auto sgpr = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  int cbr[R], cnt[R], idx[R];
  const Entry* e[R];
  // head entry of every list, the one after it, and the one after that: the entry a peek looks at was requested a whole step earlier (with
  // one entry of look-ahead the scalar load sat between the commit of a step and the peek of the next: a trip to L2 per step and wave)
  uint32_t h_a[R], h_b[R], h_w[R], n_a[R], n_b[R], n_w[R], f_a[R], f_b[R], f_w[R];
  bool any = false;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    cbr[r] = __builtin_amdgcn_readfirstlane(gr[r]);
    cnt[r] = 0, idx[r] = 0;
    e[r] = entries;
    h_a[r] = h_b[r] = h_w[r] = n_a[r] = n_b[r] = n_w[r] = f_a[r] = f_b[r] = f_w[r] = 0;
    if (cbr[r] >= 0) {
      any = true;
      cnt[r] = __builtin_amdgcn_readfirstlane(descs[cbr[r]].prod_cnt);
      const int64_t ps = descs[cbr[r]].prod_start;
      e[r] = entries + (((int64_t)__builtin_amdgcn_readfirstlane((int)(ps >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)ps));
      if (cnt[r] > 0) {
        h_a[r] = sgpr(e[r][0].a_lo), h_b[r] = sgpr(e[r][0].b_lo), h_w[r] = sgpr(e[r][0].w);
        const int i1 = cnt[r] > 1 ? 1 : 0, i2 = cnt[r] > 2 ? 2 : cnt[r] - 1;
        n_a[r] = sgpr(e[r][i1].a_lo), n_b[r] = sgpr(e[r][i1].b_lo), n_w[r] = sgpr(e[r][i1].w);
        f_a[r] = sgpr(e[r][i2].a_lo), f_b[r] = sgpr(e[r][i2].b_lo), f_w[r] = sgpr(e[r][i2].w);
      }
    }
  }
  if (!any) return;
  f32x16 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[r][q] = 0.0f;
  const int i = lane & 31, h = lane >> 5;
  const int voff = lane * 16;
  int waddr[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const int el = (c * 64 + lane) * 4;
    waddr[c] = (el / K) * PB + (el % K);
  }
  const float* rb = lds_b + (i < N ? i : N - 1) * PB + KH * h;
  const int a_voff = ((i < M ? i : M - 1) + KH * h * M) * 4;
  u32x4 sb[CB];
  float an[KH];
  auto issue_b = [&](uint32_t lo, uint32_t hi) {
    const uint64_t b_off = (uint64_t)lo | ((uint64_t)hi << 32);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + b_off), 0, K * N * 4, 0x00020000);
#pragma unroll
    for (int c = 0; c < CB; ++c) sb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff + c * 1024, 0, 0);
  };
  auto issue_a = [&](uint32_t lo, uint32_t w) {
    const uint64_t a_off = (uint64_t)lo | ((uint64_t)((w >> 16) & 0xffu) << 32);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + a_off), 0, M * K * 4, 0x00020000);
#pragma unroll
    for (int q = 0; q < KH; ++q) an[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsa, a_voff + q * M * 4, 0, 0));
  };
  // ordering key of an entry: its B offset as (bits 32-39, low word); NONE_HI marks a list that has ended
  constexpr uint32_t NONE_HI = 0xffffffffu;
  // the step AFTER the current one, looked up without touching the lists: the lists of the current step (mask `adv`) are seen one entry
  // further (their n_*), the others at their head.  Returns the smallest B offset (hi, lo), the lists that have it, and the A entry
  // (low word, w) of the first of them.
  auto peek = [&](unsigned adv, uint32_t& bhi, uint32_t& blo, unsigned& mask, uint32_t& a_lo, uint32_t& a_w) {
    bhi = NONE_HI, blo = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool ad = (adv >> r) & 1u;
      const bool live = (ad ? idx[r] + 1 : idx[r]) < cnt[r];
      const uint32_t kb = ad ? n_b[r] : h_b[r], kw = (ad ? n_w[r] : h_w[r]) >> 24;
      const bool less = live && (kw < bhi || (kw == bhi && kb < blo));
      bhi = less ? kw : bhi;
      blo = less ? kb : blo;
    }
    mask = 0, a_lo = 0, a_w = 0;
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
      const bool ad = (adv >> r) & 1u;
      const bool live = (ad ? idx[r] + 1 : idx[r]) < cnt[r];
      const uint32_t kb = ad ? n_b[r] : h_b[r], kwf = ad ? n_w[r] : h_w[r];
      const bool on = live && (kwf >> 24) == bhi && kb == blo;
      mask |= on ? (1u << r) : 0u;
      a_lo = on ? (ad ? n_a[r] : h_a[r]) : a_lo;   // (descending r: the lowest list of the step wins)
      a_w = on ? kwf : a_w;
    }
  };
  uint32_t bhi, blo, nhi = NONE_HI, nlo = 0, fa_lo, fa_w;
  unsigned mcur, mnext = 0u;
  peek(0u, bhi, blo, mcur, fa_lo, fa_w);
  if (bhi != NONE_HI) {
    issue_b(blo, bhi);
    issue_a(fa_lo, fa_w);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 bq[Q];
    // ONE product per trip (the loads of the next product's A have a single place in the code: issued from inside the R branches they
    // landed in R different register sets and were copied into the common one in the middle of the MFMA burst, waiting for them there)
    unsigned todo = mcur;   // lists of the current step that still have their product to do
    bool fresh = true;      // first product of a step
    for (;;) {
      if (fresh) {
        // the step after this one (its B block is requested as soon as this step's block has left the staging registers)
        peek(mcur, nhi, nlo, mnext, fa_lo, fa_w);
        // this step's B block: registers -> LDS (as it is stored: k contiguous), fragments read back
#pragma unroll
        for (int c = 0; c < CB; ++c) *reinterpret_cast<u32x4*>(lds_b + waddr[c]) = sb[c];
        if (nhi != NONE_HI) issue_b(nlo, nhi);
#pragma unroll
        for (int q = 0; q < Q; ++q) bq[q] = *reinterpret_cast<const f32x4*>(rb + 4 * q);
      }
      const int r = __builtin_ctz(todo);
      todo &= todo - 1u;
      // the product after this one: the next list of this step (still at its head), else the first list of the next step
      uint32_t x_lo = fa_lo, x_w = fa_w;
      const bool more = todo != 0u || mnext != 0u;
      if (todo != 0u) {
        const int q1 = __builtin_ctz(todo);
#pragma unroll
        for (int q = 0; q < R; ++q)
          if (q == q1) x_lo = h_a[q], x_w = h_w[q];
      }
      float ac[KH];
#pragma unroll
      for (int q = 0; q < KH; ++q) ac[q] = an[q];
      if (more) issue_a(x_lo, x_w);
#define DBCSR_GROUP_BURST(R_)                                                                                      \
  case R_:                                                                                                         \
    if constexpr (R_ < R) {                                                                                        \
      _Pragma("unroll") for (int q = 0; q < KH; ++q) acc[R_ < R ? R_ : 0] =                                        \
          __builtin_amdgcn_mfma_f32_32x32x2f32(bq[q >> 2][q & 3], ac[q], acc[R_ < R ? R_ : 0], 0, 0, 0);          \
    }                                                                                                              \
    break;
      switch (r) {
        DBCSR_GROUP_BURST(0)
        DBCSR_GROUP_BURST(1)
        DBCSR_GROUP_BURST(2)
        DBCSR_GROUP_BURST(3)
        default: break;
      }
#undef DBCSR_GROUP_BURST
      fresh = todo == 0u;
      if (fresh) {
        if (nhi == NONE_HI) break;
        // commit: the lists of this step move on; their next-but-one entries are requested now and looked at by the next peek
#pragma unroll
        for (int q = 0; q < R; ++q)
          if ((mcur >> q) & 1u) {
            ++idx[q];
            h_a[q] = n_a[q], h_b[q] = n_b[q], h_w[q] = n_w[q];
            n_a[q] = f_a[q], n_b[q] = f_b[q], n_w[q] = f_w[q];
            const int i3 = idx[q] + 2 < cnt[q] ? idx[q] + 2 : cnt[q] - 1;
            f_a[q] = sgpr(e[q][i3].a_lo), f_b[q] = sgpr(e[q][i3].b_lo), f_w[q] = sgpr(e[q][i3].w);
          }
        bhi = nhi, blo = nlo, mcur = mnext;
        todo = mcur;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (cbr[r] < 0 || ((skip_empty & 1) && cnt[r] == 0)) continue;
    const Desc d = descs[cbr[r]];
    float* C = c_out + d.c_off;
    const bool has_in = d.cin_off >= 0;
    const float* Ci = c_in + (has_in ? d.cin_off : 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int col = (q & 3) + 8 * (q >> 2) + 4 * h;
      if (i < M && col < N) {
        float v = alpha * acc[r][q];
        if (has_in) v += beta * Ci[i + (size_t)M * col];
        C[i + (size_t)M * col] = v;
      }
    }
  }
}


template <int S_, int R_>
static void launch_group_f32(unsigned nwg, hipStream_t st, const Desc* descs, const Entry* entries, const float* a, const float* b, float* c,
                             const float* ci, float alpha, float beta, int skip_empty, const int* groups, GroupGeom G) {
  hipLaunchKernelGGL((mm_numeric_f32_group<S_, S_, S_, R_>), dim3(nwg), dim3(256), (size_t)4 * f32d_wave_floats(S_) * sizeof(float), st, descs, entries, a, b,
                     c, ci, alpha, beta, skip_empty, groups, G);
}

int group_f32_launch(int S, int R, unsigned nwg, hipStream_t st, const Desc* descs, const Entry* entries, const float* a, const float* b, float* c,
                     const float* ci, float alpha, float beta, int skip_empty, const int* groups, GroupGeom G) {
#define DBCSR_GROUP_CASE(S_, R_) \
  case S_ * 8 + R_: launch_group_f32<S_, R_>(nwg, st, descs, entries, a, b, c, ci, alpha, beta, skip_empty, groups, G); return 0;
  switch (S * 8 + R) {
    DBCSR_GROUP_CASE(16, 2) DBCSR_GROUP_CASE(16, 3) DBCSR_GROUP_CASE(16, 4)
    DBCSR_GROUP_CASE(24, 2) DBCSR_GROUP_CASE(24, 3) DBCSR_GROUP_CASE(24, 4)
    DBCSR_GROUP_CASE(32, 2) DBCSR_GROUP_CASE(32, 3) DBCSR_GROUP_CASE(32, 4)
    default: return 1;
  }
#undef DBCSR_GROUP_CASE
}

}  // namespace dbcsr_amd
