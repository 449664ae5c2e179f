// mm_group64.h -- fp64 exact-size kernel in which a wave owns R C blocks of ONE block column and shares B among them (round 6)
// The kernels live in mm_group64.hip, a translation unit of its own; mm_engine.hip builds the tables and launches through the functions below.
//
// Why: with one wave per C block every block product pulls one B block (4.2 KB at 23 x 23) through the
// L2 <-> Infinity-Cache fabric, and config 2 runs AT that fabric's ceiling (145 GB per multiply at 7.8 TB/s).  Every dataflow of rounds 2-4
// that shares operands does so BETWEEN waves through a bounded buffer and loses to the wait.  This one shares inside a wave, so nobody waits
// for anybody: the wave keeps the accumulators of the C blocks (R g .. R g + R - 1, j) -- R x 18 registers at 23 x 23 -- and walks the union of
// their product lists in ascending k; the B block (k, j) crosses the fabric (and is laid out in LDS) ONCE per k and is multiplied with every
// A block (R g + r, k) that exists: (1 - (1 - f)^R) / (f R) B blocks per product at fill f -- 0.86 / 0.82 / 0.78 at f = 0.1, R = 4 / 5 / 6.
// The reference shares its operand slabs among the threads of a block the same way (kernels/smm_acc_dnt_largeDB2.h:159-314).
//
// OUTCOME (profiles/r06_f64_group_kernel.txt): parity-green, bit-identical to the one-wave-per-block kernel, and 19-22 % SLOWER on config 2
// (22.4-23.1 ms for R = 2 ... 6 against 18.9).  The sharing works at the L1 -> L2 level (-7 % requests at R = 4), but the L2 -> fabric requests
// RISE by 13 %: between two uses of an A block the XCD streams R times as much B and touches R times as many A rows -- a reuse distance of
// ~ 2 R x 0.6 MB against a 4 MB L2.  With the non-temporal hint on the B loads the A rows do stay (125 GB over the fabric instead of 163) but those
// loads bypass the Infinity Cache too and the kernel becomes HBM-bound (27.6 ms).  The kernel lives in the LAB build.
//
// Differences to the fp32 form of round 5 (mm_group.hip), which lost to its one-wave-per-block kernel:
//   * the union is NOT formed on the fly: a small kernel merges the R lists of every (group, column) once per plan into ONE list of 16-byte
//     records (A offset, B offset, slot r, "first product of a step") -- so the kernel can look THREE products ahead, which is what the
//     software pipeline below needs (on-the-fly merging looks one step ahead at most); the records reach the wave 64 at a time with one vector
//     load and are picked with v_readlane: no scalar load in the loop, whose out-of-order return would turn every LDS wait into lgkmcnt(0);
//   * two waves per SIMD (R accumulator sets cost occupancy), so the latency the other waves used to hide is hidden inside the wave: the A block
//     of product q + 3 and the B block of its step are requested while product q is multiplied (two register sets each, chosen by the parity
//     of q -- static after unrolling the product loop by two: every trip issues the same five A and five B loads, a trip whose product opens no
//     step issues its B loads through an empty descriptor, so all `vmcnt` distances are compile-time constants); the operands of product q + 1
//     are copied to the OTHER halves of the wave's LDS slice (A and B double-buffered) under the MFMAs of product q;
//   * products whose inner block has another size (the tail block of 32768 = 1424 x 23 + 16) do not enter the merged lists: they are summed
//     after the loop straight from global memory, as in the one-wave-per-block kernel -- the summation order per C block, and with it every
//     bit of the result, is that kernel's.
// Launch order as the fp32 form: XCD x takes the row groups g = x (mod 8) -- the A rows of a group live in its L2 --, sweeps the columns of a
// panel of B and moves to the next group.  C blocks of another size than S x S are left to a second launch of the one-wave-per-block kernel.
#ifndef DBCSR_AMD_MM_GROUP64_H
#define DBCSR_AMD_MM_GROUP64_H

#include "common.h"
#include "mm_types.h"

namespace dbcsr_amd {

struct GroupGeom {
  int nbc;   // block columns of C
  int ng;    // row groups
  int ngx;   // row groups per XCD (the largest share: XCDs with fewer find empty positions)
  int pw;    // columns per panel
  int np;    // panels
};

struct GEntry {  // one product of a merged list, 16 bytes
  uint32_t a_lo, b_lo;  // low 32 bits of the element offsets of the A / B block
  uint32_t w;           // bits 0-7 / 8-15: bits 32-39 of the A / B offset; bits 16-18: slot r of the C block; bit 24: first product of its step
  uint32_t pad;
};

struct GWork {  // one (group, column): its merged list, its C blocks and its first product in ONE 64-byte read
  int64_t start;        // first record in gentries
  int32_t cnt;          // records (products whose inner block has the dominant size)
  int32_t cb[8];        // C block of slot r, -1: none
  uint32_t a_lo, b_lo, w;  // record 0 (undefined when cnt == 0)
  uint32_t pad;
};

// flag_dev[0] (zeroed by the caller) becomes non-zero when some block of the matrix lies before its predecessor in index order
void group_check_ascending(hipStream_t st, const int64_t* blk_p, int64_t nblks, int* flag_dev);
// groups[(g * nbc + j) * R + r] = index of the C block (R g + r, j) when it exists and is S x S, else -1
void group_build_table(hipStream_t st, const int* c_row_p, const int* c_col_i, const Desc* descs, int nbr, int nbc, int R, int S, int* groups);
// cnt[t] = upper bound of the merged list's length of (group, column) t: the sum of its blocks' product counts
void group64_count(hipStream_t st, const int* groups, const Desc* descs, int64_t ngj, int R, int* cnt);
// merged lists: gentries[gstart[t] ...] and the records gwork[t] with their true lengths (products whose k extent is not K are left out)
void group64_merge(hipStream_t st, const int* groups, const Desc* descs, const Entry* entries, int64_t ngj, int R, int K, const int64_t* gstart,
                   GWork* gwork, GEntry* gentries);
// 0 = launched, 1 = no kernel for this (S, R)
int group64_launch(int S, int R, hipStream_t st, const Desc* descs, const Entry* entries, const double* a, const double* b, double* c, const double* ci,
                   double alpha, double beta, int skip_empty, int has_tail, const GWork* gwork, const GEntry* gentries, GroupGeom G);
bool group64_has_kernel(int S, int R);

}  // namespace dbcsr_amd
#endif
