// mm_mid.hip -- the instantiations of mm_numeric_f64_mid (one wave per C block, operands in slabs through 6-13 KB of LDS) and their launcher.
// Slabs of 16 inner indices while the block has at most 8 units of 4 x 4 per dimension (32 x 32: 130 registers, three waves per SIMD), of 8 above
// (40 x 40 with 16 would need 40 staging registers more than three waves per SIMD leave).
#include "mm_mid.h"

#include "smm_core.h"
#include "mm_numeric_f64_mid.h"

namespace dbcsr_amd {

// Which shapes it should take.  Measured (profiles/r06_slab_kernel.txt):
//  * a dimension beyond 32 (9 / 10 units): the alternative is the workgroup kernel, which it beats (33^3 0.29 -> 0.36, 40^3 0.46 -> 0.55 of the fp64 peak);
//  * uniform blocks of 25 ... 32: the exact-size kernels win or tie (28^3 0.43 against 0.39, 30^3 0.49 / 0.42; 32^3 0.55 / 0.53 once hot<32,32,32>
//    stages with a padded pitch) -- not taken;
//  * the (32, 32), (32, 23), (23, 32) CLASSES of a mixed-size multiply with few products per C block (config 3: 3.6): the class kernels stage whole
//    blocks in 15-18 KB per wave, two waves per SIMD; with the slab kernel on these three classes config 3 takes 7.15 instead of 7.6 ms -- taken
//    (DBCSR_AMD_MM_MID=3: only (32, 32)).
//  * a dimension of 41 ... 48 (11 / 12 units; session r06_24): a tie with the workgroup kernel on cubes (41^3 0.40 / 0.40, 44^3 0.48 / 0.47, 45^3 0.45 / 0.46,
//    48^3 0.60 / 0.56 -- 0.31 / 0.33 at 5 % fill), a clear win when the other dimension is at most 40 (48 x 36 x 23: 0.51 against 0.43) -- taken then,
//    and for multiples of 4 in both (no padding inside the units).
bool mid_f64_serves(int m, int n, int class_mode) {
  const int rb = (m + 3) / 4, cb = (n + 3) / 4;
  const int lo = rb < cb ? rb : cb, hi = rb < cb ? cb : rb;
  if (lo < 6 || hi > 12 || hi < 8) return false;
  if (hi >= 11) return lo <= 10 || (m % 4 == 0 && n % 4 == 0);
  if (hi >= 9) return true;
  return class_mode > 0 && (lo == 8 || (lo == 6 && class_mode != 3));
}

bool launch_mid_f64(int rb, int cb, bool other_sizes, unsigned npos, hipStream_t st, const Desc* descs, int64_t nblk, const Entry* entries,
                    const double* a_data, const double* b_data, double* c_out, const double* c_in, double alpha, double beta, int skip_empty,
                    const int* order, const Work* work, int max_units, double* norms) {
  if (npos == 0 || rb < 6 || cb < 6 || rb > 12 || cb > 12) return false;
  // the shape that covers every block of the multiply (the second launch; the only one when the dominant size is that shape)
  const int fb = (max_units > 10 || rb > 10 || cb > 10) ? 12 : 10;
  const bool single = rb == fb && cb == fb;
  const int flags = (skip_empty & 1) | (single ? 32 : 0);
  switch (rb * 16 + cb) {
#define DBCSR_MID_CASE(A_, B_)                                                                                                                   \
  case A_ * 16 + B_: {                                                                                                                           \
    constexpr int KSL = (A_ <= 8 && B_ <= 8) ? 16 : 8;                                                                                           \
    hipLaunchKernelGGL((mm_numeric_f64_mid<A_, B_, KSL>), dim3(npos), dim3(64), (size_t)mid_lds_bytes((A_ + 1) / 2, (B_ + 1) / 2, KSL), st, descs, nblk, \
                       entries, a_data, b_data, c_out, c_in, alpha, beta, flags, order, work, norms);                                            \
  } break;
    DBCSR_MID_CASE(6, 8) DBCSR_MID_CASE(6, 9) DBCSR_MID_CASE(6, 10) DBCSR_MID_CASE(6, 11) DBCSR_MID_CASE(6, 12)
    DBCSR_MID_CASE(7, 9) DBCSR_MID_CASE(7, 10) DBCSR_MID_CASE(7, 11) DBCSR_MID_CASE(7, 12)
    DBCSR_MID_CASE(8, 6) DBCSR_MID_CASE(8, 8) DBCSR_MID_CASE(8, 9) DBCSR_MID_CASE(8, 10) DBCSR_MID_CASE(8, 11) DBCSR_MID_CASE(8, 12)
    DBCSR_MID_CASE(9, 6) DBCSR_MID_CASE(9, 7) DBCSR_MID_CASE(9, 8) DBCSR_MID_CASE(9, 9) DBCSR_MID_CASE(9, 10) DBCSR_MID_CASE(9, 11) DBCSR_MID_CASE(9, 12)
    DBCSR_MID_CASE(10, 6) DBCSR_MID_CASE(10, 7) DBCSR_MID_CASE(10, 8) DBCSR_MID_CASE(10, 9) DBCSR_MID_CASE(10, 10) DBCSR_MID_CASE(10, 11) DBCSR_MID_CASE(10, 12)
    DBCSR_MID_CASE(11, 6) DBCSR_MID_CASE(11, 7) DBCSR_MID_CASE(11, 8) DBCSR_MID_CASE(11, 9) DBCSR_MID_CASE(11, 10) DBCSR_MID_CASE(11, 11) DBCSR_MID_CASE(11, 12)
    DBCSR_MID_CASE(12, 6) DBCSR_MID_CASE(12, 7) DBCSR_MID_CASE(12, 8) DBCSR_MID_CASE(12, 9) DBCSR_MID_CASE(12, 10) DBCSR_MID_CASE(12, 11) DBCSR_MID_CASE(12, 12)
#undef DBCSR_MID_CASE
    default: return false;
  }
  if (!single && other_sizes) {
    const int f2 = (skip_empty & 1) | 16 | (rb << 8) | (cb << 12);
    if (fb == 12)
      hipLaunchKernelGGL((mm_numeric_f64_mid<12, 12, 8>), dim3(npos), dim3(64), (size_t)mid_lds_bytes(6, 6, 8), st, descs, nblk, entries, a_data, b_data, c_out,
                         c_in, alpha, beta, f2, order, work, norms);
    else
      hipLaunchKernelGGL((mm_numeric_f64_mid<10, 10, 8>), dim3(npos), dim3(64), (size_t)mid_lds_bytes(5, 5, 8), st, descs, nblk, entries, a_data, b_data, c_out,
                         c_in, alpha, beta, f2, order, work, norms);
  }
  return true;
}

}  // namespace dbcsr_amd
