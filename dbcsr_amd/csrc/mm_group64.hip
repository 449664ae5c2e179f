// mm_group64.hip -- the fp64 group dataflow (mm_group64.h): table / merged-list builders and the kernel.  A translation unit of its own:
// the kernel multiplies into one of R accumulator sets under a wave-uniform switch and is compiled with the options mm_tile.hip needs for
// the same reason (Makefile).
#include "mm_group64.h"
#include "smm_core.h"
#include <type_traits>
#include <cstdlib>

namespace dbcsr_amd {

static inline dim3 g64_grid_for(int64_t nthreads) { return dim3((unsigned)((nthreads + 255) / 256)); }

// groups[(g * nbc + j) * R + r] = index of the C block (R g + r, j) when it exists and is M x N, else -1
__global__ void __launch_bounds__(256) build_groups(const int* __restrict__ c_row_p, const int* __restrict__ c_col_i, const Desc* __restrict__ descs,
                                                    int nbr, int nbc, int R, int M, int N, int* __restrict__ groups) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ng = (nbr + R - 1) / R;
  if (tid >= ng * nbc * R) return;
  const int r = (int)(tid % R);
  const int64_t gj = tid / R;
  const int j = (int)(gj % nbc), i = (int)(gj / nbc) * R + r;
  int cb = -1;
  if (i < nbr) {
    int lo = c_row_p[i], hi = c_row_p[i + 1];  // binary search of column j in the sorted row
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (c_col_i[mid] < j)
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo < c_row_p[i + 1] && c_col_i[lo] == j && descs[lo].m == M && descs[lo].n == N) cb = lo;
  }
  groups[tid] = cb;
}

// flag[0] != 0 afterwards: some block of the matrix lies before its predecessor in index order
__global__ void __launch_bounds__(256) blk_p_not_ascending(const int64_t* __restrict__ blk_p, int64_t nblks, int* __restrict__ flag) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b + 1 < nblks && blk_p[b + 1] <= blk_p[b]) flag[0] = 1;
}

void group_check_ascending(hipStream_t st, const int64_t* blk_p, int64_t nblks, int* flag_dev) {
  if (nblks > 1) hipLaunchKernelGGL(blk_p_not_ascending, g64_grid_for(nblks), dim3(256), 0, st, blk_p, nblks, flag_dev);
}

void group_build_table(hipStream_t st, const int* c_row_p, const int* c_col_i, const Desc* descs, int nbr, int nbc, int R, int S, int* groups) {
  const int64_t ng = (nbr + R - 1) / R;
  hipLaunchKernelGGL(build_groups, g64_grid_for(ng * nbc * R), dim3(256), 0, st, c_row_p, c_col_i, descs, nbr, nbc, R, S, S, groups);
}

// ---- merged lists ------------------------------------------------------------------------------------------------------------------
constexpr int G64_MAXR = 8;

__global__ void __launch_bounds__(256) group64_count_k(const int* __restrict__ groups, const Desc* __restrict__ descs, int64_t ngj, int R,
                                                       int* __restrict__ cnt) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ngj) return;
  int s = 0;
  for (int r = 0; r < R; ++r) {
    const int cb = groups[t * R + r];
    if (cb >= 0) s += descs[cb].prod_cnt;
  }
  cnt[t] = s;
}

// One thread per (group, column): an R-way merge of the blocks' lists by B offset (every list ascends in k, and with B's blocks in index
// order the offsets of one column ascend with k: group_check_ascending).  Lists are short (14 products at config 2) and the whole pass moves
// 0.8 GB there; it runs once per plan.
__global__ void __launch_bounds__(256) group64_merge_k(const int* __restrict__ groups, const Desc* __restrict__ descs, const Entry* __restrict__ entries,
                                                       int64_t ngj, int R, int K, const int64_t* __restrict__ gstart, GWork* __restrict__ gwork,
                                                       GEntry* __restrict__ gentries) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ngj) return;
  const Entry* e[G64_MAXR];
  int cnt[G64_MAXR], idx[G64_MAXR];
  uint64_t key[G64_MAXR];  // B offset of the list's head (~0: the list has ended)
  Entry head[G64_MAXR];
  auto advance = [&](int r) {   // move list r to its next product of inner size K
    key[r] = ~0ull;
    while (idx[r] < cnt[r]) {
      const Entry x = e[r][idx[r]];
      if (x.ks() == K) {
        head[r] = x;
        key[r] = x.b_off();
        break;
      }
      ++idx[r];
    }
  };
#pragma unroll
  for (int r = 0; r < G64_MAXR; ++r) {
    cnt[r] = 0, idx[r] = 0, e[r] = entries, key[r] = ~0ull;
    if (r < R) {
      const int cb = groups[t * R + r];
      if (cb >= 0) {
        cnt[r] = descs[cb].prod_cnt;
        e[r] = entries + descs[cb].prod_start;
        advance(r);
      }
    }
  }
  GEntry* out = gentries + gstart[t];
  int n = 0;
  for (;;) {
    uint64_t kmin = ~0ull;
#pragma unroll
    for (int r = 0; r < G64_MAXR; ++r) kmin = key[r] < kmin ? key[r] : kmin;
    if (kmin == ~0ull) break;
    bool first = true;
#pragma unroll
    for (int r = 0; r < G64_MAXR; ++r)
      if (key[r] == kmin) {
        GEntry o;
        o.a_lo = head[r].a_lo, o.b_lo = head[r].b_lo;
        o.w = ((head[r].w >> 16) & 0xffffu) | ((uint32_t)r << 16) | (first ? (1u << 24) : 0u);
        o.pad = 0;
        out[n++] = o;
        first = false;
        ++idx[r];
        advance(r);
      }
  }
  GWork w;
  w.start = gstart[t];
  w.cnt = n;
#pragma unroll
  for (int r = 0; r < G64_MAXR; ++r) w.cb[r] = r < R ? groups[t * R + r] : -1;
  w.a_lo = n > 0 ? out[0].a_lo : 0u, w.b_lo = n > 0 ? out[0].b_lo : 0u, w.w = n > 0 ? out[0].w : 0u;
  w.pad = 0;
  gwork[t] = w;
}

void group64_count(hipStream_t st, const int* groups, const Desc* descs, int64_t ngj, int R, int* cnt) {
  hipLaunchKernelGGL(group64_count_k, g64_grid_for(ngj), dim3(256), 0, st, groups, descs, ngj, R, cnt);
}

void group64_merge(hipStream_t st, const int* groups, const Desc* descs, const Entry* entries, int64_t ngj, int R, int K, const int64_t* gstart,
                   GWork* gwork, GEntry* gentries) {
  hipLaunchKernelGGL(group64_merge_k, g64_grid_for(ngj), dim3(256), 0, st, groups, descs, entries, ngj, R, K, gstart, gwork, gentries);
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------------------
template <int S>
struct G64 {
  static constexpr int MA = (S + 7) / 8, KS = (S + 3) / 4, K4 = 4 * KS;
  static constexpr int A_IMG = S * K4 * 8;                 // A as an S x K4 column-major image, columns S .. K4 - 1 exact zeros
  static constexpr int B_IMG = (S * S * 8 + 15) & ~15;     // B as stored
  static constexpr int CA = (A_IMG + 1023) / 1024, CB = (B_IMG + 1023) / 1024, CC = (S * S * 8 + 1023) / 1024;
  static constexpr int DUMP = 2 * A_IMG + 2 * B_IMG;       // 1 KiB nobody reads: where the lanes past an image's end put their piece of its last 1 KiB
  static constexpr int LDS_BYTES = DUMP + 1024;            // per wave: A double-buffered (parity of the product), B double-buffered (parity of the step)
};

// Software pipeline of one wave (q = product, in merged-list order; PAR = q & 1, static after unrolling by two):
//   trip q:  LDS <- registers: A(q + 1) into A image 1 - PAR, and -- if product q + 1 opens a step -- its B into the other B image
//            registers <- memory: A(q + 3) and, if product q + 3 opens a step, its B (else the same five loads through an EMPTY descriptor:
//            they return zeros and touch no memory, and every trip issues the same ten loads -- all vmcnt distances are constants)
//            54 MFMAs of product q from A image PAR and the current B image into accumulator set r(q)   [wave-uniform switch]
// The list records come in batches of 64 with ONE vector load (lane l holds record 64 b + l) and are picked with v_readlane: a scalar load
// in the loop would turn every LDS wait of the burst into lgkmcnt(0) INCLUDING that load (SMEM returns out of order) -- the first version
// did that and ran 23 ms whatever R was (gpurun_out/r06_s01).
// BNT: the B loads carry the non-temporal hint (DBCSR_AMD_MM_GROUP_BNT=1; an experiment: does a streamed B leave the group's A rows in L2?)
template <int S, int R, bool BNT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
mm_numeric_f64_group(const Desc* __restrict__ descs, const Entry* __restrict__ entries, const double* __restrict__ a_data,
                     const double* __restrict__ b_data, double* __restrict__ c_out, const double* __restrict__ c_in, double alpha, double beta,
                     int skip_empty, int has_tail, const GWork* __restrict__ gwork, const GEntry* __restrict__ gentries, GroupGeom G) {
  static_assert(S >= 9 && S <= 32 && R >= 2 && R <= G64_MAXR, "mm_numeric_f64_group: shape");
  typedef G64<S> X;
  constexpr int MA = X::MA, NC = X::MA, KS = X::KS, A_IMG = X::A_IMG, B_IMG = X::B_IMG, CA = X::CA, CB = X::CB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  // position of this wave in its XCD's stream -> (panel, row group, column)
  const int xcd = (int)(blockIdx.x & 7u);
  const int s = (int)(blockIdx.x >> 3);  // (the host keeps ngx * nbc below 2^28)
  const int full = G.ngx * G.pw;
  int p = s / full;
  if (p > G.np - 1) p = G.np - 1;
  const int rem = s - p * full;
  const int pwl = p == G.np - 1 ? G.nbc - p * G.pw : G.pw;
  const int gl = rem / pwl, j = p * G.pw + rem % pwl;
  const int g = xcd + 8 * gl;
  if (gl >= G.ngx || g >= G.ng) return;
  const int64_t t = (int64_t)g * G.nbc + j;
  auto sgpr = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  // the group's record: list, C blocks, first product -- ONE read, and the first operands can be requested at once
  const GWork gw = gwork[t];
  const int cnt = (int)sgpr((uint32_t)gw.cnt);
  int cbr[R];
  bool any = false;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    cbr[r] = (int)sgpr((uint32_t)gw.cb[r]);
    any |= cbr[r] >= 0;
  }
  if (!any) return;
  const uint64_t gs = ((uint64_t)sgpr((uint32_t)((uint64_t)gw.start >> 32)) << 32) | sgpr((uint32_t)gw.start);
  const GEntry* ge = gentries + gs;

  char* lds_a = smem;                 // two A images
  char* lds_b = smem + 2 * A_IMG;     // two B images
  const LaneMap L(lane);
  const int voff = lane * 16;
  double acc[R][MA][NC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[r][a][c] = 0.0;
  // fragment addresses (image 0 of A; image `bsel` of B)
  const double* pa[MA];
  const double* pb[NC];
  const double* pbt[NC];  // last k step when S is not a multiple of 4: lanes past the end read element (0, col) (A's padding is zero)
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    int row = 8 * a + L.rowl;
    row = row < S ? row : S - 1;
    pa[a] = reinterpret_cast<const double*>(lds_a) + row + S * L.kq;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = 8 * c + L.coll;
    col = col < S ? col : S - 1;
    pb[c] = reinterpret_cast<const double*>(lds_b) + L.kq + S * col;
    const int kt = 4 * (KS - 1) + L.kq;
    pbt[c] = reinterpret_cast<const double*>(lds_b) + (kt < S ? kt : 0) + S * col;
  }
  // where a lane stores its 16 bytes of the LAST 1 KiB piece of an image: inside the image, or (past its end) in the dump area -- no
  // lane-divergent branch around a store, and the compiler has no reason to sink a load into one
  constexpr bool A_TAIL = (A_IMG & 1023) != 0, B_TAIL = (B_IMG & 1023) != 0;
  int a_last[2], b_last[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_last[i] = (!A_TAIL || voff < A_IMG - 1024 * (CA - 1)) ? i * A_IMG + 1024 * (CA - 1) + voff : X::DUMP + voff;
    b_last[i] = (!B_TAIL || voff < B_IMG - 1024 * (CB - 1)) ? 2 * A_IMG + i * B_IMG + 1024 * (CB - 1) + voff : X::DUMP + voff;
  }
  u32x4 ra[2][CA], rb[2][CB];
  // an empty descriptor (no records): every lane out of range -- the loads return zeros and touch no memory
  auto issue_a = [&](u32x4 (&dst)[CA], uint64_t a_off, bool real) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + a_off), 0, real ? S * S * 8 : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < CA; ++c) dst[c] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, c * 1024, 0);
  };
  auto issue_b = [&](u32x4 (&dst)[CB], uint64_t b_off, bool real) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + b_off), 0, real ? S * S * 8 : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < CB; ++c) dst[c] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, c * 1024, BNT ? 2 : 0);
  };
  auto store_a = [&](const u32x4 (&src)[CA], int img) {
#pragma unroll
    for (int c = 0; c < CA - 1; ++c) *reinterpret_cast<u32x4*>(lds_a + img * A_IMG + c * 1024 + voff) = src[c];
    *reinterpret_cast<u32x4*>(smem + a_last[img]) = src[CA - 1];
  };
  auto store_b = [&](const u32x4 (&src)[CB], int img) {
#pragma unroll
    for (int c = 0; c < CB - 1; ++c) *reinterpret_cast<u32x4*>(lds_b + img * B_IMG + c * 1024 + voff) = src[c];
    *reinterpret_cast<u32x4*>(smem + (img ? b_last[1] : b_last[0])) = src[CB - 1];
  };
  struct SE {  // a list record in scalar registers
    uint32_t a_lo, b_lo, w;
    __device__ __forceinline__ uint64_t a_off() const { return (uint64_t)a_lo | ((uint64_t)(w & 0xffu) << 32); }
    __device__ __forceinline__ uint64_t b_off() const { return (uint64_t)b_lo | ((uint64_t)((w >> 8) & 0xffu) << 32); }
    __device__ __forceinline__ int r() const { return (int)((w >> 16) & 7u); }
    __device__ __forceinline__ bool first() const { return (w >> 24) & 1u; }
  };
  if (cnt > 0) {
    SE e0;
    e0.a_lo = sgpr(gw.a_lo), e0.b_lo = sgpr(gw.b_lo), e0.w = sgpr(gw.w);
    // product 0's operands first, then the records: batch b = records 64 b ... 64 b + 63, one per lane (past the list's end: zeros)
    issue_b(rb[0], e0.b_off(), true);
    issue_a(ra[0], e0.a_off(), true);
    const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)ge, 0, cnt * (int)sizeof(GEntry), 0x00020000);
    // (plain locals that no lambda captures: captured by reference, the conditional copy `batch <- next batch` became a phi of two stack slots
    // and a scratch load -- counted by vmcnt, waited for with vmcnt(0) -- sat in every trip)
    int bat0, bat1, bat2, nxt0, nxt1, nxt2;
    {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsl, voff, 0, 0);
      bat0 = nxt0 = (int)v[0], bat1 = nxt1 = (int)v[1], bat2 = nxt2 = (int)v[2];
    }
#define DBCSR_G64_PICK(E_, QQ_)                                                  \
  do {                                                                          \
    const int l_ = (QQ_) & 63;                                                  \
    (E_).a_lo = (uint32_t)__builtin_amdgcn_readlane(bat0, l_);                  \
    (E_).b_lo = (uint32_t)__builtin_amdgcn_readlane(bat1, l_);                  \
    (E_).w = (uint32_t)__builtin_amdgcn_readlane(bat2, l_);                     \
  } while (0)
// the record of product QQ_ = q + 3 for the trip that follows; lists of more than 64 products change batches here (the next batch is requested 48 trips ahead)
#define DBCSR_G64_NEXT_RECORD(QQ_)                                                                                 \
  do {                                                                                                            \
    const int q3_ = (QQ_);                                                                                        \
    if ((q3_ & 63) == 16 && q3_ + 48 < cnt) {                                                                     \
      const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(rsl, voff, ((q3_ >> 6) + 1) * 1024, 0);              \
      nxt0 = (int)v_[0], nxt1 = (int)v_[1], nxt2 = (int)v_[2];                                                    \
    }                                                                                                             \
    if ((q3_ & 63) == 0) { /* (the empty asm keeps this a BRANCH: as a select it would wait for the next batch's load in every trip) */ \
      asm volatile("" ::: "memory");                                                                              \
      bat0 = nxt0, bat1 = nxt1, bat2 = nxt2;                                                                      \
    }                                                                                                             \
    DBCSR_G64_PICK(e3, q3_);                                                                                      \
  } while (0)
    SE e1, e2, e3;
    DBCSR_G64_PICK(e1, 1);
    DBCSR_G64_PICK(e2, 2);
    DBCSR_G64_PICK(e3, 3);
    int q = 0;
    int bsel = 0;  // the B image of the current step
    // prologue: product 0 into the LDS images 0, products 1 and 2 in flight
    issue_a(ra[1], e1.a_off(), 1 < cnt);
    issue_b(rb[1], e1.b_off(), 1 < cnt && e1.first());
    store_a(ra[0], 0);
    store_b(rb[0], 0);
    issue_a(ra[0], e2.a_off(), 2 < cnt);
    issue_b(rb[0], e2.b_off(), 2 < cnt && e2.first());
    const double* pbx[NC];
    const double* pbtx[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) pbx[c] = pb[c], pbtx[c] = pbt[c];
    auto burst = [&](double (&ac)[MA][NC], int par) {
#pragma unroll
      for (int s4 = 0; s4 < KS; ++s4) {
        double av[MA], bv[NC];
#pragma unroll
        for (int a = 0; a < MA; ++a) av[a] = pa[a][par * (A_IMG / 8) + s4 * 4 * S];
#pragma unroll
        for (int c = 0; c < NC; ++c) bv[c] = (s4 == KS - 1 && (S & 3)) ? pbtx[c][0] : pbx[c][4 * s4];
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
          for (int c = 0; c < NC; ++c) ac[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[a], bv[c], ac[a][c], 0, 0, 0);
      }
    };
    auto trip = [&](auto par_c) {
      constexpr int PAR = decltype(par_c)::value;
      // (1) the operands of product q + 1 (requested two trips ago) go to the other LDS images
      store_a(ra[1 - PAR], 1 - PAR);
      const bool step_next = q + 1 < cnt && e1.first();
      if (step_next) store_b(rb[1 - PAR], bsel ^ 1);
      // (2) the operands of product q + 3 are requested into the registers just freed
      issue_a(ra[1 - PAR], e3.a_off(), q + 3 < cnt);
      issue_b(rb[1 - PAR], e3.b_off(), q + 3 < cnt && e3.first());
      // (4) product q
      switch (e0.r()) {
#define DBCSR_G64_CASE(R_)                                  \
  case R_:                                                  \
    if constexpr (R_ < R) burst(acc[R_ < R ? R_ : 0], PAR); \
    break;
        DBCSR_G64_CASE(0) DBCSR_G64_CASE(1) DBCSR_G64_CASE(2) DBCSR_G64_CASE(3)
        DBCSR_G64_CASE(4) DBCSR_G64_CASE(5) DBCSR_G64_CASE(6) DBCSR_G64_CASE(7)
#undef DBCSR_G64_CASE
        default: break;
      }
      // (5) on to product q + 1 (the record of ITS third successor is picked by the loop below)
      if (step_next) {
        bsel ^= 1;
#pragma unroll
        for (int c = 0; c < NC; ++c) pbx[c] = pb[c] + bsel * (B_IMG / 8), pbtx[c] = pbt[c] + bsel * (B_IMG / 8);
      }
      e0 = e1, e1 = e2, e2 = e3;
      ++q;
    };
    for (;;) {
      trip(std::integral_constant<int, 0>());
      if (q >= cnt) break;
      DBCSR_G64_NEXT_RECORD(q + 3);
      trip(std::integral_constant<int, 1>());
      if (q >= cnt) break;
      DBCSR_G64_NEXT_RECORD(q + 3);
    }
#undef DBCSR_G64_NEXT_RECORD
#undef DBCSR_G64_PICK
  }
  // products with an inner block of another size (the tail block column of A): straight from global memory, after the others -- the
  // order of the one-wave-per-block kernel
  if (has_tail) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (cbr[r] < 0) continue;
      const Desc d = descs[cbr[r]];
      const Entry* e = entries + d.prod_start;
      for (int pp = 0; pp < d.prod_cnt; ++pp) {
        const Entry ep = e[pp];
        if (ep.ks() != S) block_product_f64<MA, NC, false>(acc[r], a_data + ep.a_off(), b_data + ep.b_off(), S, S, ep.ks(), L);
      }
    }
  }
  // the R C blocks leave through LDS in whole 1 KiB pieces with the streaming hint (as cblock_f64_exact); their descriptors and C_in
  // blocks are requested together, ahead of the first block's turn
  constexpr int CC = X::CC;
  double* lds_c = reinterpret_cast<double*>(lds_a);
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  Desc dr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) dr[r] = descs[cbr[r] >= 0 ? cbr[r] : 0];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (cbr[r] < 0) continue;
    const Desc& d = dr[r];
    const int64_t c_off = (int64_t)(((uint64_t)sgpr((uint32_t)((uint64_t)d.c_off >> 32)) << 32) | sgpr((uint32_t)d.c_off));
    const int64_t cin_off = (int64_t)(((uint64_t)sgpr((uint32_t)((uint64_t)d.cin_off >> 32)) << 32) | sgpr((uint32_t)d.cin_off));
    const int pc = (int)sgpr((uint32_t)d.prod_cnt);
    if ((skip_empty & 1) && pc == 0) continue;
    const bool has_in = cin_off >= 0;
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(c_out + c_off), 0, S * S * 8, 0x00020000);
    u32x4 ci[CC];
    if (has_in) {
      const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(c_in + cin_off), 0, S * S * 8, 0x00020000);
#pragma unroll
      for (int c = 0; c < CC; ++c) ci[c] = __builtin_amdgcn_raw_buffer_load_b128(rsi, voff, c * 1024, 0);
    }
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int row = 8 * a + L.rowd, col = 8 * c + L.coll;
        if (row < S && col < S) lds_c[row + S * col] = alpha * acc[r][a][c];
      }
    if (has_in) {
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        f64x2 v = *reinterpret_cast<const f64x2*>(lds_a + c * 1024 + voff);
        const f64x2 w = __builtin_bit_cast(f64x2, ci[c]);
        v[0] += beta * w[0];
        v[1] += beta * w[1];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsc, voff + c * 1024, 0, 2);
      }
    } else {
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(lds_a + c * 1024 + voff);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsc, voff + c * 1024, 0, 2);
      }
    }
  }
}

template <int S_, int R_>
static void launch_group64(unsigned nwg, hipStream_t st, const Desc* descs, const Entry* entries, const double* a, const double* b, double* c,
                           const double* ci, double alpha, double beta, int skip_empty, int has_tail, const GWork* gwork, const GEntry* gentries,
                           GroupGeom G) {
  static const bool bnt = getenv("DBCSR_AMD_MM_GROUP_BNT") != nullptr && atoi(getenv("DBCSR_AMD_MM_GROUP_BNT")) != 0;
  if (bnt)
    hipLaunchKernelGGL((mm_numeric_f64_group<S_, R_, true>), dim3(nwg), dim3(64), (size_t)G64<S_>::LDS_BYTES, st, descs, entries, a, b, c, ci, alpha, beta,
                       skip_empty, has_tail, gwork, gentries, G);
  else
    hipLaunchKernelGGL((mm_numeric_f64_group<S_, R_, false>), dim3(nwg), dim3(64), (size_t)G64<S_>::LDS_BYTES, st, descs, entries, a, b, c, ci, alpha, beta,
                       skip_empty, has_tail, gwork, gentries, G);
}

#define DBCSR_G64_SHAPES(X) X(16, 2) X(16, 3) X(16, 4) X(23, 2) X(23, 3) X(23, 4) X(23, 5) X(23, 6)

bool group64_has_kernel(int S, int R) {
#define DBCSR_G64_HAS(S_, R_) \
  if (S == S_ && R == R_) return true;
  DBCSR_G64_SHAPES(DBCSR_G64_HAS)
#undef DBCSR_G64_HAS
  return false;
}

int group64_launch(int S, int R, hipStream_t st, const Desc* descs, const Entry* entries, const double* a, const double* b, double* c, const double* ci,
                   double alpha, double beta, int skip_empty, int has_tail, const GWork* gwork, const GEntry* gentries, GroupGeom G) {
  const unsigned nwg = 8u * (unsigned)((int64_t)G.ngx * G.nbc);
#define DBCSR_G64_LAUNCH(S_, R_)                                                                                                              \
  if (S == S_ && R == R_) {                                                                                                                   \
    launch_group64<S_, R_>(nwg, st, descs, entries, a, b, c, ci, alpha, beta, skip_empty, has_tail, gwork, gentries, G);          \
    return 0;                                                                                                                                 \
  }
  DBCSR_G64_SHAPES(DBCSR_G64_LAUNCH)
#undef DBCSR_G64_LAUNCH
  return 1;
}

}  // namespace dbcsr_amd
