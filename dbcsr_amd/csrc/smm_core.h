// smm_core.h -- per-wavefront small-block GEMM cores for gfx950 (CDNA4).
//
// fp64: built on v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per
// instruction).  Measured on MI355X (profiles/r01_ubench_fp64_mfma.txt): this
// form sustains 76.5 TFLOP/s chip-wide (the fp64 peak), whereas
// v_mfma_f64_16x16x4_f64 reaches only 45-65 TFLOP/s; and a 4-wide tiling pads
// a 23x23x23 block to 24x24x24 (88 % useful) instead of 32x32x24 (49 %).
//
// Operand layout of v_mfma_f64_4x4x4_4b_f64 (probed on hardware, same file):
//   lane l:  kq = l >> 4,  blk = (l >> 2) & 3,  x = l & 3
//   A operand  = A_blk[i = x][k = kq]
//   B operand  = B_blk[k = kq][j = x]
//   D (1 f64)  = D_blk[i = kq][j = x]          (i.e. row index sits in l >> 4)
// One instruction is used as a 2x2 arrangement of 4x4 tiles: blk = 2p + q
// handles rows 8a + 4p + [0,4) and columns 8c + 4q + [0,4) of the C block, so
// one A register (row group pair a) and one B register (column group pair c)
// feed it, and an (8*MA) x (8*NC) C block needs MA + NC operand registers and
// MA*NC instructions per 4-deep k step.
//
// Blocks are column-major as DBCSR stores them: A is m x k,
// C is m x n, B is k x n (BT = false) or, after libsmm_acc_transpose,
// n x k (BT = true).
#ifndef DBCSR_AMD_SMM_CORE_H
#define DBCSR_AMD_SMM_CORE_H

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace dbcsr_amd {

struct LaneMap {
  int kq;    // k index inside a 4-deep step (operands) == row index inside a tile (result)
  int rowl;  // 4p + x : row inside an 8-row group pair (A operand)
  int coll;  // 4q + x : column inside an 8-column group pair (B operand, result)
  int rowd;  // 4p + kq: row inside an 8-row group pair (result)
  __device__ __forceinline__ explicit LaneMap(int lane) {
    kq = lane >> 4;
    const int p = (lane >> 3) & 1, q = (lane >> 2) & 1, x = lane & 3;
    rowl = 4 * p + x;
    coll = 4 * q + x;
    rowd = 4 * p + kq;
  }
};

// acc[a][c] += A(m x k) * B(k x n) restricted to the C tile starting at
// (row0, col0) of extent (8*MA) x (8*NC); row0 = col0 = 0 when the block fits.
// Rows/columns beyond m/n are fed from clamped (valid) addresses: they only
// pollute result rows/columns that are never stored.  The k tail is zeroed
// exactly.
template <int MA, int NC, bool BT>
__device__ __forceinline__ void block_product_f64(double (&acc)[MA][NC], const double* __restrict__ A,
                                                  const double* __restrict__ B, int m, int n, int k, const LaneMap& L,
                                                  int row0 = 0, int col0 = 0) {
  int aoff[MA], boff[NC];
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    int row = row0 + 8 * a + L.rowl;
    row = row < m ? row : m - 1;
    aoff[a] = row + m * L.kq;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = col0 + 8 * c + L.coll;
    col = col < n ? col : n - 1;
    boff[c] = BT ? (col + n * L.kq) : (L.kq + k * col);
  }
  const int astep = 4 * m, bstep = BT ? 4 * n : 4;
  const int kfull = k & ~3;
  int kb = 0;
  for (; kb < kfull; kb += 4) {
    double av[MA], bv[NC];
#pragma unroll
    for (int a = 0; a < MA; ++a) av[a] = A[aoff[a]];
#pragma unroll
    for (int c = 0; c < NC; ++c) bv[c] = B[boff[c]];
#pragma unroll
    for (int a = 0; a < MA; ++a) aoff[a] += astep;
#pragma unroll
    for (int c = 0; c < NC; ++c) boff[c] += bstep;
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
  }
  if (kb < k) {  // k tail: lanes whose k index is past the end contribute exact zeros
    const bool kv = (kb + L.kq) < k;
    double av[MA], bv[NC];
#pragma unroll
    for (int a = 0; a < MA; ++a) {
      const double v = A[kv ? aoff[a] : 0];
      av[a] = kv ? v : 0.0;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double v = B[kv ? boff[c] : 0];
      bv[c] = kv ? v : 0.0;
    }
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
  }
}

// Same product with both operands already staged in LDS: A as an m x k4 column-major
// block whose columns k..k4-1 (k4 = k rounded up to 4) are EXACT ZEROS, B as stored
// (k x n column-major).  Lanes whose k index is past the end read A's zero padding and
// a clamped (valid, finite) B element, so they contribute exact zeros without any
// select on loaded values -- which lets the operand fetches of step s+1 stay in flight
// under the MFMAs of step s (two-stage software pipeline).
template <int MA, int NC, bool BT = false>
__device__ __forceinline__ void block_product_f64_lds(double (&acc)[MA][NC], const double* lds_a, const double* lds_b, int m,
                                                      int n, int k, const LaneMap& L) {
  int aoff[MA], boff[NC];
#pragma unroll
  for (int a = 0; a < MA; ++a) {
    int row = 8 * a + L.rowl;
    row = row < m ? row : m - 1;
    aoff[a] = row + m * L.kq;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = 8 * c + L.coll;
    col = col < n ? col : n - 1;
    boff[c] = BT ? (col + n * L.kq) : (L.kq + k * col);  // BT: B staged as libsmm_acc_transpose leaves it (n x k)
  }
  const int nsteps = (k + 3) >> 2;
  const int astep = 4 * m;
  const int bstep = BT ? 4 * n : 4;
  const int klast = k - 1 - L.kq;  // 4*s <= klast  <=>  this lane's k index is inside the block
  auto fetch = [&](int s, double (&av)[MA], double (&bv)[NC]) {
#pragma unroll
    for (int a = 0; a < MA; ++a) av[a] = lds_a[aoff[a] + s * astep];
    const int bs = 4 * s <= klast ? s * bstep : (BT ? -n * L.kq : -L.kq);  // past the end: element (k = 0, col), always valid
#pragma unroll
    for (int c = 0; c < NC; ++c) bv[c] = lds_b[boff[c] + bs];
  };
  auto mma = [&](const double (&av)[MA], const double (&bv)[NC]) {
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
  };
  double av0[MA], bv0[NC], av1[MA], bv1[NC];
  fetch(0, av0, bv0);
  int s = 0;
  for (; s + 2 <= nsteps; s += 2) {
    fetch(s + 1, av1, bv1);
    mma(av0, bv0);
    if (s + 2 < nsteps) fetch(s + 2, av0, bv0);
    mma(av1, bv1);
  }
  if (s < nsteps) mma(av0, bv0);
}

// fp32 core on v_mfma_f32_32x32x2_f32: one instruction covers a whole block of
// up to 32 x 32 for 2 k.  lane l: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31];
// result register r (0..15): col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
typedef float f32x16 __attribute__((ext_vector_type(16)));

// SWAP: the operands change places in the instruction -- acc then holds the TRANSPOSED tile (register r of lane l: column
// n = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), row m = l & 31), the layout of the kernels that feed A straight from global memory.
template <bool BT, bool SWAP = false>
__device__ __forceinline__ void block_product_f32(f32x16& acc, const float* __restrict__ A, const float* __restrict__ B, int m,
                                                  int n, int k, int lane, int row0 = 0, int col0 = 0) {
  const int i = lane & 31, kh = lane >> 5;
  const int row = (row0 + i) < m ? (row0 + i) : m - 1;
  const int col = (col0 + i) < n ? (col0 + i) : n - 1;
  int aoff = row + m * kh;
  int boff = BT ? (col + n * kh) : (kh + k * col);
  const int astep = 2 * m, bstep = BT ? 2 * n : 2;
  const int kfull = k & ~1;
  int kb = 0;
  for (; kb < kfull; kb += 2) {
    const float av = A[aoff], bv = B[boff];
    aoff += astep;
    boff += bstep;
    acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  }
  if (kb < k) {
    const bool kv = kh == 0;
    const float a0 = A[kv ? aoff : 0], b0 = B[kv ? boff : 0];
    const float az = kv ? a0 : 0.0f, bz = kv ? b0 : 0.0f;
    acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(bz, az, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(az, bz, acc, 0, 0, 0);
  }
}

}  // namespace dbcsr_amd
#endif
