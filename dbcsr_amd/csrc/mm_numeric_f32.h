// mm_numeric_f32.h -- fp32 block-product kernels
// Part of the device-resident multiply engine: included by mm_engine.hip (one translation unit), in this order:
// mm_workspace.h, mm_symbolic.h, mm_numeric_f64.h, mm_numeric_f32.h, mm_aux.h.
#ifndef DBCSR_AMD_MM_NUMERIC_F32_H
#define DBCSR_AMD_MM_NUMERIC_F32_H

namespace dbcsr_amd {

// ---- fp32, LDS-staged (blocks up to 32 x 32; BASELINE config 5) -----------------------------------------
// One v_mfma_f32_32x32x2_f32 covers the whole C block for 2 k.  A (m x k, column-major) is copied to LDS as
// is: its fragment (lane = row, two k per instruction) reads 32 consecutive floats.  B is stored k x n with k
// contiguous, but its fragment wants n across lanes at a fixed k -- 32 lanes 128 B apart would all hit one LDS
// bank -- so B is written to LDS TRANSPOSED with a row pitch of 33 floats (Bt[j + 33 kk]); the staging write
// computes (kk, j) per element with a multiply-shift division by the runtime k.
constexpr int F32_CH = 4;            // 1 KiB chunks: 4 x 256 floats >= 32 x 32
constexpr int F32_LDN = 33;          // pitch of the transposed B image
constexpr int F32_A_FLOATS = 1024 + 64, F32_BT_FLOATS = F32_LDN * 32 + 31;
constexpr int F32_WAVE_FLOATS = F32_A_FLOATS + ((F32_BT_FLOATS + 3) & ~3);

__device__ __forceinline__ void cblock_f32_lds(const Desc& d, const Entry* __restrict__ entries, const float* __restrict__ a_data,
                                               const float* __restrict__ b_data, float* __restrict__ c_out,
                                               const float* __restrict__ c_in, float alpha, float beta, int lane, float* lds_a,
                                               float* lds_bt) {
  constexpr int CH = F32_CH, LDN = F32_LDN;
  const int m = d.m, n = d.n, cnt = d.prod_cnt;
  const Entry* e = entries + d.prod_start;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  u32x4 ra[CH], rb[CH];
  const int voff = lane * 16;
  auto issue = [&](int p) {
    const int ks = e[p].ks();
    const int abytes = __builtin_amdgcn_readfirstlane(m * ks * 4), bbytes = __builtin_amdgcn_readfirstlane(ks * n * 4);  // scalar on purpose, see cblock_f64_lds
    const int nca = __builtin_amdgcn_readfirstlane((m * (ks + 1) * 4 + 1023) >> 10), ncb = (bbytes + 1023) >> 10;  // A: one zero column of k padding
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + e[p].a_off()), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + e[p].b_off()), 0, bbytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < nca) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < ncb) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
  };
  if (cnt > 0) issue(0);
  const int i = lane & 31, kh = lane >> 5;
  const int arow = i < m ? i : m - 1, bcol = i < n ? i : n - 1;
  for (int p = 0; p < cnt; ++p) {
    const int ks = e[p].ks();
    const int kn = ks * n;
    const int nca = __builtin_amdgcn_readfirstlane((m * (ks + 1) * 4 + 1023) >> 10), ncb = __builtin_amdgcn_readfirstlane((kn * 4 + 1023) >> 10);
    const unsigned inv = (65536u + (unsigned)ks - 1u) / (unsigned)ks;  // j = (e * inv) >> 16 == e / ks for e < 2048, ks <= 32
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < nca) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(lds_a) + c * 1024 + voff) = ra[c];
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < ncb) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const unsigned el = (unsigned)((c * 64 + lane) * 4 + t);
          const unsigned j = (el * inv) >> 16, kk = el - j * (unsigned)ks;
          if ((int)el < kn) lds_bt[j + LDN * kk] = __uint_as_float(rb[c][t]);
        }
      }
    if (p + 1 < cnt) issue(p + 1);
    // multiply: lane (i, kh) feeds A[i][2s + kh] and B[2s + kh][i]; the odd-k tail reads A's zero padding column
    const int nsteps = (ks + 1) >> 1;
    int aoff = arow + m * kh;
    for (int s2 = 0; s2 < nsteps; ++s2) {
      const int kk = 2 * s2 + kh;
      const float av = lds_a[aoff];
      const float bv = lds_bt[bcol + LDN * (kk < ks ? kk : ks - 1)];
      aoff += 2 * m;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  float* C = c_out + d.c_off;
  const bool has_in = d.cin_off >= 0;
  const float* Ci = c_in + (has_in ? d.cin_off : 0);
  const int col = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < m && col < n) {
      float v = alpha * acc[r];
      if (has_in) v += beta * Ci[row + (size_t)m * col];
      C[row + (size_t)m * col] = v;
    }
  }
}

// Exact-size fp32 variant (see cblock_f64_exact): with M, N, K known at compile time the transposed LDS image of B needs no
// per-element division (the generic kernel spends 208 VALU + 140 SALU instructions per 32^3 product next to 16 MFMAs,
// MFMA pipe 42 % busy): the LDS address of every staged element is a per-wave constant.
template <int M, int N, int K>
__device__ __forceinline__ void cblock_f32_exact(const Desc& d, const Entry* __restrict__ entries, const float* __restrict__ a_data,
                                                 const float* __restrict__ b_data, float* __restrict__ c_out,
                                                 const float* __restrict__ c_in, float alpha, float beta, int lane, float* lds_a,
                                                 float* lds_bt) {
  constexpr int LDN = F32_LDN;
  constexpr int KS2 = (K + 1) / 2, KP = 2 * KS2;                       // k steps of 2; A is zero-padded to KP columns
  constexpr int CA = (M * KP * 4 + 1023) / 1024, CB = (K * N * 4 + 1023) / 1024;
  constexpr int DUMMY = F32_BT_FLOATS;                                 // LDS slot that swallows the staging lanes past the block end
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const Entry* e = entries + d.prod_start;
  const int cnt = d.prod_cnt;
  u32x4 ra[CA], rb[CB];
  const int voff = lane * 16;
  int baddr[CB][4];  // where element t of chunk c of this lane goes in the transposed image: constants of the wave
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int el = (c * 64 + lane) * 4 + t;
      const int j = el / K, kk = el - j * K;
      baddr[c][t] = el < K * N ? j + LDN * kk : DUMMY;
    }
  const int i = lane & 31, kh = lane >> 5;
  const float* pa = lds_a + (i < M ? i : M - 1) + M * kh;
  const float* pb = lds_bt + (i < N ? i : N - 1) + LDN * kh;
  const float* pbt = lds_bt + (i < N ? i : N - 1) + LDN * ((K & 1) && kh ? K - 1 : 2 * (KS2 - 1) + kh);  // last step of an odd K
  auto issue = [&](uint64_t a_off, uint64_t b_off) {
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + a_off), 0, M * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + b_off), 0, K * N * 4, 0x00020000);
#pragma unroll
    for (int c = 0; c < CA; ++c) ra[c] = __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, c * 1024, 0);
#pragma unroll
    for (int c = 0; c < CB; ++c) rb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, c * 1024, 0);
  };
  int i0 = 0;
  Entry e0 = e[0];
  while (i0 < cnt && e0.ks() != K) {
    ++i0;
    e0 = e[i0 < cnt ? i0 : cnt - 1];
  }
  int i1 = i0 + 1;
  Entry e1 = e[i1 < cnt ? i1 : cnt - 1];
  if (i0 < cnt) issue(e0.a_off(), e0.b_off());
  while (i0 < cnt) {
#pragma unroll
    for (int c = 0; c < CA; ++c) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(lds_a) + c * 1024 + voff) = ra[c];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) lds_bt[baddr[c][t]] = __uint_as_float(rb[c][t]);
    while (i1 < cnt && e1.ks() != K) {
      ++i1;
      e1 = e[i1 < cnt ? i1 : cnt - 1];
    }
    if (i1 < cnt) issue(e1.a_off(), e1.b_off());
    const Entry e2 = e[i1 + 1 < cnt ? i1 + 1 : cnt - 1];
#pragma unroll
    for (int s2 = 0; s2 < KS2; ++s2) {
      const float av = pa[s2 * 2 * M];
      const float bv = (s2 == KS2 - 1) ? pbt[0] : pb[s2 * 2 * LDN];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    i0 = i1;
    e0 = e1;
    i1 = i1 + 1;
    e1 = e2;
  }
  for (int p = 0; p < cnt; ++p) {
    const Entry ep = e[p];
    if (ep.ks() != K) block_product_f32<false>(acc, a_data + ep.a_off(), b_data + ep.b_off(), M, N, ep.ks(), lane);
  }
  float* C = c_out + d.c_off;
  const bool has_in = d.cin_off >= 0;
  const float* Ci = c_in + (has_in ? d.cin_off : 0);
  const int col = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < M && col < N) {
      float v = alpha * acc[r];
      if (has_in) v += beta * Ci[row + (size_t)M * col];
      C[row + (size_t)M * col] = v;
    }
  }
}

// ---- exact-size fp32, A straight from global memory (round 5) --------------------------------------------------
// What bounded cblock_f32_exact (0.60 of the fp32 peak also on cache-resident operands, profiles/r04_config5_kpass_views.txt) is the
// LDS: per 32^3 product 4 ds_write_b128 + 16 ds_write_b32 + 32 ds_read_b32 = 180 LDS-pipe cycles per wave (MI355X_MICROARCH.md, LDS
// table) against the 1024 cycles of its 16 MFMAs -- 0.7 of the LDS of a CU whose four SIMDs all multiply.  Here the wave computes the
// TRANSPOSED tile, C^T = B^T A^T:
//   * first MFMA operand (32 x 2, lane = output row) = B^T: lane (n, h) wants B[k][n] for the k of its half.  B is stored k x n with k
//     contiguous, so the block goes to LDS AS IT IS -- 16-byte stores, row pitch K + 4 floats -- and ONE ds_read_b128 hands a lane four
//     consecutive k of its column: 4 ds_write_b128 + 4 ds_read_b128 = 68 LDS-pipe cycles per product, no transposition, no scalar
//     LDS traffic.  (Pitch K + 4: the 16 lanes of each ds_read_b128 lane group hit the 16 distinct four-bank groups for K = 16, 24, 32.)
//   * second operand (2 x 32, lane = output column) = A^T: lane (m, h) wants A[m][k].  A is m x k with m contiguous: for a fixed k the
//     32 lanes of a half read 128 consecutive bytes -- a plain dword load per k step, straight into the operand register, no LDS.
//   * the order of the k sum is free: step j multiplies k = j (lanes 0-31) and k = K/2 + j (lanes 32-63), which is what makes a lane's
//     four B values of one ds_read_b128 four successive steps' operands.
// acc: register r of lane l = C[m = l & 31][n = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]: for a fixed r a half-wave writes 128
// consecutive bytes of C.  K must be a multiple of 8 (a half's k range a multiple of 4); the other sizes keep cblock_f32_exact.
// (F32D_ROWS, f32d_pitch, f32d_wave_floats: mm_types.h, shared with the lab's group form)

template <int M, int N, int K>
__device__ __forceinline__ void cblock_f32_direct(const Desc& d, const Entry* __restrict__ entries, const float* __restrict__ a_data,
                                                  const float* __restrict__ b_data, float* __restrict__ c_out,
                                                  const float* __restrict__ c_in, float alpha, float beta, int lane, float* lds_b) {
  static_assert(K % 8 == 0 && K >= 8 && K <= 32 && M <= 32 && N <= 32, "cblock_f32_direct: K a multiple of 8, blocks of at most 32");
  constexpr int PB = f32d_pitch(K), KH = K / 2, Q = KH / 4;
  constexpr int CB = (K * N * 4 + 1023) / 1024;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const Entry* e = entries + d.prod_start;
  const int cnt = d.prod_cnt;
  const int i = lane & 31, h = lane >> 5;
  const int voff = lane * 16;
  // where this lane's 16 bytes of chunk c go in the image: four consecutive k of one column (K is a multiple of 4)
  int waddr[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const int el = (c * 64 + lane) * 4;
    waddr[c] = (el / K) * PB + (el % K);
  }
  const float* rb = lds_b + (i < N ? i : N - 1) * PB + KH * h;         // this lane's column of B, its half of k
  const int a_voff = ((i < M ? i : M - 1) + KH * h * M) * 4;           // this lane's row of A, its half of k
  u32x4 sb[CB];     // B block of the NEXT product on its way to LDS
  float an[KH];     // A operand of the NEXT product
  // (two register sets that change roles from product to product -- no copy -- were tried in round 5: the twice-unrolled loop needed 116
  //  registers instead of 84)
  auto issue = [&](uint64_t a_off, uint64_t b_off) {
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_data + b_off), 0, K * N * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_data + a_off), 0, M * K * 4, 0x00020000);
#pragma unroll
    for (int c = 0; c < CB; ++c) sb[c] = __builtin_amdgcn_raw_buffer_load_b128(rsb, voff + c * 1024, 0, 0);
#pragma unroll
    for (int j = 0; j < KH; ++j) an[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsa, a_voff + j * M * 4, 0, 0));
  };
  int i0 = 0;
  Entry e0 = e[0];
  while (i0 < cnt && e0.ks() != K) {
    ++i0;
    e0 = e[i0 < cnt ? i0 : cnt - 1];
  }
  int i1 = i0 + 1;
  Entry e1 = e[i1 < cnt ? i1 : cnt - 1];
  if (i0 < cnt) issue(e0.a_off(), e0.b_off());
  while (i0 < cnt) {
#pragma unroll
    for (int c = 0; c < CB; ++c) *reinterpret_cast<u32x4*>(lds_b + waddr[c]) = sb[c];
    float ac[KH];
#pragma unroll
    for (int j = 0; j < KH; ++j) ac[j] = an[j];
    while (i1 < cnt && e1.ks() != K) {
      ++i1;
      e1 = e[i1 < cnt ? i1 : cnt - 1];
    }
    if (i1 < cnt) issue(e1.a_off(), e1.b_off());
    const Entry e2 = e[i1 + 1 < cnt ? i1 + 1 : cnt - 1];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 bq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) bq[q] = *reinterpret_cast<const f32x4*>(rb + 4 * q);
#pragma unroll
    for (int j = 0; j < KH; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[j >> 2][j & 3], ac[j], acc, 0, 0, 0);
    i0 = i1;
    e0 = e1;
    i1 = i1 + 1;
    e1 = e2;
  }
  for (int p = 0; p < cnt; ++p) {
    const Entry ep = e[p];
    if (ep.ks() != K) block_product_f32<false, true>(acc, a_data + ep.a_off(), b_data + ep.b_off(), M, N, ep.ks(), lane);
  }
  float* C = c_out + d.c_off;
  const bool has_in = d.cin_off >= 0;
  const float* Ci = c_in + (has_in ? d.cin_off : 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int col = (r & 3) + 8 * (r >> 2) + 4 * h;
    if (i < M && col < N) {
      float v = alpha * acc[r];
      if (has_in) v += beta * Ci[i + (size_t)M * col];
      C[i + (size_t)M * col] = v;
    }
  }
}

static inline size_t f32_lds_bytes(int wg_waves) { return ((size_t)wg_waves * F32_WAVE_FLOATS + 4) * sizeof(float); }
#define DBCSR_F32_KERNEL_HEAD                                                                          \
  extern __shared__ __attribute__((aligned(16))) char smem_raw_[]; /* f32_lds_bytes(waves per workgroup) */ \
  float* smem = reinterpret_cast<float*>(smem_raw_);                                                   \
  const int lane = threadIdx.x & 63;                                                                   \
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                             \
  const int wg = xcd_remap(blockIdx.x, gridDim.x);                                                     \
  const int64_t pos = (int64_t)wg * (int)(blockDim.x >> 6) + wid;                                      \
  const int64_t cb = order[pos];                                                                       \
  if (cb < 0 || cb >= nblk) return;                                                                    \
  const Desc d = descs[cb];                                                                            \
  if ((skip_empty & 1) && d.prod_cnt == 0) return;                                                     \
  float* lds_a = smem + (size_t)wid * F32_WAVE_FLOATS;                                                 \
  float* lds_bt = lds_a + F32_A_FLOATS;

__global__ void __launch_bounds__(256) mm_numeric_f32_lds(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                          const float* __restrict__ a_data, const float* __restrict__ b_data,
                                                          float* __restrict__ c_out, const float* __restrict__ c_in, float alpha,
                                                          float beta, int skip_empty, const int* __restrict__ order) {
  DBCSR_F32_KERNEL_HEAD
  cblock_f32_lds(d, entries, a_data, b_data, c_out, c_in, alpha, beta, lane, lds_a, lds_bt);
}

template <int M, int N, int K>
__global__ void __launch_bounds__(256) mm_numeric_f32_hot(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                          const float* __restrict__ a_data, const float* __restrict__ b_data,
                                                          float* __restrict__ c_out, const float* __restrict__ c_in, float alpha,
                                                          float beta, int skip_empty, const int* __restrict__ order) {
  DBCSR_F32_KERNEL_HEAD
  if (d.m == M && d.n == N)
    cblock_f32_exact<M, N, K>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, lane, lds_a, lds_bt);
  else
    cblock_f32_lds(d, entries, a_data, b_data, c_out, c_in, alpha, beta, lane, lds_a, lds_bt);
}

// the direct form: per wave only the image of B (f32d_wave_floats(K) floats); blocks of another size take the generic staged path in the
// same slice when it is large enough for it, else the plain global-memory product
template <int M, int N, int K>
__global__ void __launch_bounds__(256) mm_numeric_f32_direct(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                             const float* __restrict__ a_data, const float* __restrict__ b_data,
                                                             float* __restrict__ c_out, const float* __restrict__ c_in, float alpha,
                                                             float beta, int skip_empty, const int* __restrict__ order) {
  DBCSR_F32_KERNEL_HEAD
  if ((skip_empty & 2) && d.m == M && d.n == N) return;  // the group kernel (mm_numeric_f32_group.h) computed the blocks of the dominant size
  if (d.m == M && d.n == N)
    cblock_f32_direct<M, N, K>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, lane, lds_a);
  else
    cblock_f32_lds(d, entries, a_data, b_data, c_out, c_in, alpha, beta, lane, lds_a, lds_bt);
}

// the same when EVERY C block of the launch is M x N (no tail block row / column: config 5): the workgroup's LDS is the B images of its waves and
// nothing else -- 4.5 KB per wave instead of the 8.7 KB the staged fall-back of the kernel above needs, which held the CU at 16 waves
template <int M, int N, int K>
__global__ void __launch_bounds__(256) mm_numeric_f32_direct_slim(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                                  const float* __restrict__ a_data, const float* __restrict__ b_data,
                                                                  float* __restrict__ c_out, const float* __restrict__ c_in, float alpha,
                                                                  float beta, int skip_empty, const int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw_[];   // blockDim / 64 * f32d_wave_floats(K) floats
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t pos = (int64_t)wg * (int)(blockDim.x >> 6) + wid;
  const int64_t cb = order[pos];
  if (cb < 0 || cb >= nblk) return;
  const Desc d = descs[cb];
  if ((skip_empty & 1) && d.prod_cnt == 0) return;
  if (d.m != M || d.n != N) return;   // (never: the host launches this form only when all C blocks have the dominant size)
  cblock_f32_direct<M, N, K>(d, entries, a_data, b_data, c_out, c_in, alpha, beta, lane, reinterpret_cast<float*>(smem_raw_) + (size_t)wid * f32d_wave_floats(K));
}

__global__ void __launch_bounds__(256) mm_numeric_f32(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                      const float* __restrict__ a_data, const float* __restrict__ b_data,
                                                      float* __restrict__ c_out, const float* __restrict__ c_in, float alpha,
                                                      float beta, int skip_empty) {
  const int lane = threadIdx.x & 63;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t cb = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6));
  if (cb >= nblk) return;
  const Desc d = descs[cb];
  if ((skip_empty & 1) && d.prod_cnt == 0) return;
  const int m = d.m, n = d.n;
  const Entry* e = entries + d.prod_start;
  const bool has_in = d.cin_off >= 0;
  for (int row0 = 0; row0 < m; row0 += 32)
    for (int col0 = 0; col0 < n; col0 += 32) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      for (int p = 0; p < d.prod_cnt; ++p)
        block_product_f32<false>(acc, a_data + e[p].a_off(), b_data + e[p].b_off(), m, n, e[p].ks(), lane, row0, col0);
      float* C = c_out + d.c_off;
      const float* Ci = c_in + (has_in ? d.cin_off : 0);
      const int col = col0 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < m && col < n) {
          float v = alpha * acc[r];
          if (has_in) v += beta * Ci[row + (size_t)m * col];
          C[row + (size_t)m * col] = v;
        }
      }
    }
}

}  // namespace dbcsr_amd
#endif
