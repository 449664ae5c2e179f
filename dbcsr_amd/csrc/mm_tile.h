// mm_tile.h -- XCD-wide C tiles held in registers: the second dataflow of the fp64 block-product engine.
//
// The one-wave-per-C-block kernels (mm_engine.hip) read one A and one B block per block product; the A block-row of a C row
// stays in its XCD's L2 but every B block crosses the L2 <-> Infinity-Cache fabric, and at 10 % fill that stream (123 GB per
// multiply of BASELINE config 2) is what bounds them.  Here the C blocks of an I x J TILE are accumulated at the same
// time, so that an A block (i, k) is shared by the tile's columns and a B block (k, j) by its rows: every operand block crosses
// the fabric once per tile, 10 (|I| + |J|) / (|I| |J|) blocks per product at 10 % fill (0.42 for 48 x 48) instead of 1.1.
//   * a wavefront owns a sub-tile of C blocks (3 x 3: 81 fp64 accumulators per lane, two waves per SIMD; or 4 x 3: 108, one wave per
//     SIMD -- the shapes below) and walks ONE product list, sorted by k, whose entries name the accumulator set they feed ("slot");
//   * the 256 waves of an XCD (32 CUs x 8 waves: one persistent workgroup per CU) form a TEAM that works on one 16 x 16
//     arrangement of sub-tiles (48 x 48 C blocks) at a time; sharing happens in the XCD's 4 MB L2, which only holds about 100 k
//     steps of the tile's operands, so the team moves through k together: every wave publishes the k it needs next, and a wave
//     may fetch operands for k only while k <= (minimum over the team) + W.  The protocol affects speed only: a wave that waited
//     too long switches its throttle off, and results never depend on where workgroups were placed;
//   * operands reach LDS by LDS-DMA (dma_lds.h) into a two-slot ring per wave -- no staging VGPRs (there are none to spare) --
//     and fragments are read with single ds_read_b64 (MI355X_MICROARCH.md, LDS table: a ds_read2_b64 costs 8 LDS cycles, two
//     ds_read_b64 4);
//   * block rows / columns / inner blocks of another size than the dominant (M, N, K) stay out of the tiles: their C blocks
//     are left to the exact-size kernel's launch over "the other sizes", their products with M x N blocks to tile_remainder.
// Index work (tile_* kernels): per tile the 9 descriptors and one k-sorted product list, built from the bitmaps of A (rows) and of
// B transposed (columns) with wave-wide prefix sums; integer work, bit-exact by construction, checked against the per-block
// lists in the tests.
#ifndef DBCSR_AMD_MM_TILE_H
#define DBCSR_AMD_MM_TILE_H

#include "dma_lds.h"

namespace dbcsr_amd {

// Two SHAPES of the dataflow (DBCSR_AMD_MM_TILE_SHAPE):
//   0: sub-tiles of 3 x 3 C blocks, 8 waves per workgroup (two per SIMD, 256 registers each), ring of 2 slots per wave;
//      team = 256 waves = 16 x 16 sub-tiles = 48 x 48 C blocks
//   1: sub-tiles of 4 x 3 C blocks, 4 waves per workgroup (ONE per SIMD: 108 accumulators per lane in the 512 registers), ring of
//      4 slots per wave -- three products' operands in flight behind the one being multiplied, which covers a trip over the fabric
//      and lets a wave that is admitted late keep multiplying what it already has; team = 128 waves = 8 x 16 sub-tiles = 32 x 48
//      C blocks
constexpr int kTileMaxT = 4;                   // largest sub-tile edge of any shape
constexpr int kTileMaxSlots = 16;              // C blocks per wave, at most (slot = tc * ti + tj)
constexpr int kTeamCols = 16;                  // sub-tiles per super-tile row (team waves = rows x kTeamCols)
constexpr unsigned kTileDone = 0x7fffffffu;    // progress value of a wave that needs nothing any more

struct TileEntry {  // one block product of a sub-tile, 16 bytes
  uint32_t a_lo, b_lo;  // low 32 bits of the element offsets into the A / B data areas
  uint32_t w;           // bits 0-3: slot (3 ti + tj); bits 8-15: k extent; bits 16-23 / 24-31: bits 32-39 of the A / B offset
  uint32_t k;           // inner block index (position in the team's sweep)
};

struct TileDesc {  // one sub-tile, 272 bytes
  int64_t list_start;  // first TileEntry; the n_main entries with inner size K come first (sorted by k), the n_rem others last
  int32_t n_main, n_rem;
  int64_t c_off[kTileMaxSlots];    // element offset in C_out data, -1: no C block in this slot
  int64_t cin_off[kTileMaxSlots];  // element offset in C_in data, -1: the block is new
};

struct TileGeom {
  int nfr, nfc;        // block rows / columns of the dominant size
  int nTR, nTC;        // sub-tile grid
  int nSR, nSC;        // super-tile grid
  int team_rows;       // sub-tile rows of a super-tile (team waves = team_rows * kTeamCols)
  int nseq;            // super-tiles per XCD (ceil)
  int kspan;           // progress units per super-tile (>= number of inner blocks)
  int tr, tc;          // C block rows / columns of a sub-tile (the shape's)
  int wg_waves;        // waves per workgroup = per CU (the shape's)
};

// ---- numeric kernels: mm_tile.hip (a translation unit of its own, see the Makefile) ------------------------------------------

struct TileArgs {
  const TileDesc* tdescs;
  const TileEntry* entries;
  const double* a_data;
  const double* b_data;
  double* c_out;
  const double* c_in;
  double alpha, beta;
  unsigned* prog;  // [8][256] progress of the team's waves (zeroed before the launch)
  int* flags;      // diagnostics: [0] waves that gave up waiting, [1] list mismatches (index kernels), [2] polls / 16, [3] blocked products / 16
  TileGeom G;
  int window;      // W, in inner blocks; <= 0: no throttle
  int pub_policy;  // progress stores: 0 = written through (device scope), 1 = left in the XCD's L2
  int prefetch;    // 1: the blocks of the product after the next one are pulled into L2 ahead of their DMA
  unsigned long long* times;  // [6] sums over the waves (knob bit 5): total, window waits, operand waits, multiplies, epilogues [10 ns], waves
  int knobs;       // tuning switches of the team protocol (mm_tile.hip): bits 0-2 publish quantum shift, 3 issue priority, 4 short sleep
};

// block sizes the tile kernels are built for (cubes)
#define DBCSR_AMD_TILE_SIZES(X) X(23)

// sub-tile and workgroup geometry of a shape; false: no such shape
bool tile_shape(int shape, int* tr, int* tc, int* wg_waves);
// LDS bytes of a workgroup, 0 when there is no kernel for (m, n, k)
int tile_lds_bytes(int m, int n, int k, int shape);
// the persistent tile kernel (one workgroup per CU, nwg workgroups: 8 per CU slot of an XCD) and the products with inner blocks of
// another size; 0 = launched, 1 = no kernel for this size, < 0 error
int tile_launch(int m, int n, int k, int rdv, int shape, unsigned nwg, hipStream_t st, const TileArgs& P);
int tile_launch_remainder(int m, int n, hipStream_t st, const TileGeom& G, const TileDesc* tdescs, const TileEntry* entries, const double* a_data,
                          const double* b_data, double* c_out, double alpha);

}  // namespace dbcsr_amd
#endif
