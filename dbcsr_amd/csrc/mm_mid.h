// mm_mid.h -- launcher of the one-wave slab kernels (mm_numeric_f64_mid.h), compiled in a translation unit of their own (mm_mid.hip)
#ifndef DBCSR_AMD_MM_MID_H
#define DBCSR_AMD_MM_MID_H
#include <hip/hip_runtime.h>

#include "mm_types.h"

namespace dbcsr_amd {

// C blocks of rb x cb units of 4 x 4 (6 ... 12 per dimension, the larger at least 8: 21 ... 48 rows / columns with at least one dimension above 28) on the
// launch-order positions order[0 .. npos) (work: their records, or null): one wave per block.  other_sizes: positions may hold blocks of another
// size -- a second launch of the largest shape (<10, 10> up to 40, <12, 12> up to 48) takes them.  false: no kernel for this shape (nothing was launched).
bool launch_mid_f64(int rb, int cb, bool other_sizes, unsigned npos, hipStream_t st, const Desc* descs, int64_t nblk, const Entry* entries,
                    const double* a_data, const double* b_data, double* c_out, const double* c_in, double alpha, double beta, int skip_empty,
                    const int* order, const Work* work, int max_units, double* norms);   // max_units: the largest block dimension of the launch, in units of 4;
                                                                                             // norms (may be null): every block's squared Frobenius norm as stored
// is this shape one the slab kernel should take?  class_mode: 0 = the dominant size of a multiply, 1 = an (m, n) class of a mixed-size multiply
// (3: as 1 without the classes of 21 ... 24 in one dimension); see mm_mid.hip
bool mid_f64_serves(int m, int n, int class_mode);   // (m, n: rows and columns of the dominant block / of the class)

}  // namespace dbcsr_amd
#endif
