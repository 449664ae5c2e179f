// mm_group.h -- fp32 exact-size kernel in which a wave owns R C blocks of ONE block column and shares B among them (round 5)
// The kernels live in mm_group.hip, a translation unit of its own (see there); mm_engine.hip builds the group table and launches through the
// functions below.
//
// Why (gpurun_out/r05_s02, profiles/r05_f32_*): with one wave per C block every block product pulls one B block (4 KiB at 32 x 32) through
// the L2 <-> Infinity-Cache fabric -- 16384^2, 32768^2 and config 5's four k passes all run at the same 6.3-6.4 TB/s of B blocks, whatever
// the kernel's LDS budget (the direct form of mm_numeric_f32.h cut the LDS work by 60 % and gained 0-2 %).  The reference shares its
// operand slabs among all the threads of a block (kernels/smm_acc_dnt_largeDB2.h:159-314); here, at block granularity:
//   * a wave keeps the accumulators of the C blocks (i_0 .. i_{R-1}, j) -- R x 16 registers -- and walks the UNION of their product lists
//     in ascending k: the B block (k, j) is fetched and laid out in LDS once per k and multiplied with every A block (i_r, k) that exists.
//     At fill f a B block then serves f R products on average: (1 - (1 - f)^R) / (f R) B blocks per product -- 0.74 at f = 0.2, R = 4;
//   * the A blocks come straight from global memory into the MFMA operand (as in cblock_f32_direct), one product ahead;
//   * the union is formed on the fly from the R per-block lists the symbolic phase already made: every list is ascending in k, and for one
//     column j the B offsets ascend with k when B's blocks lie in memory in index order (checked once per plan: `b_monotone`; every matrix
//     this library makes does, a host's work matrix may not -- then the one-wave-per-block kernel runs).  Scalar work only: R heads, their
//     minimum, a mask of the lists that have it.
// Launch order: XCD x takes the row groups g = x (mod 8) -- the A rows of ONE group live in its L2 --, sweeps the columns of a panel of B
// (sized for the Infinity Cache) and then moves to the next group; the position -> (group, column) map is arithmetic, `groups` holds the
// C block of every (group, column, slot) or -1.  Blocks of other sizes than M x N are left to a second launch of the one-wave-per-block
// kernel.  Needs: every inner block of size K (no k tail), K a multiple of 8.
#ifndef DBCSR_AMD_MM_GROUP_H
#define DBCSR_AMD_MM_GROUP_H

#include "common.h"
#include "mm_types.h"
#include "mm_group64.h"   // GroupGeom, group_check_ascending, group_build_table: shared with the fp64 form

namespace dbcsr_amd {

// (the image of B in LDS -- F32D_ROWS, f32d_pitch, f32d_wave_floats -- is mm_types.h's: shared with the direct form of mm_numeric_f32.h)

// 0 = launched, 1 = no kernel for this (S, R)
int group_f32_launch(int S, int R, unsigned nwg, hipStream_t st, const Desc* descs, const Entry* entries, const float* a, const float* b, float* c,
                     const float* ci, float alpha, float beta, int skip_empty, const int* groups, GroupGeom G);

}  // namespace dbcsr_amd
#endif
