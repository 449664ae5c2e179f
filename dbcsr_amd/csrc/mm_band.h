// mm_band.h -- CU-wide C tiles with the B operand shared in LDS: the third dataflow of the fp64 block-product engine.
//
// Why.  The one-wave-per-C-block kernel (mm_numeric_f64.h) stages one A and one B block per block product: at BASELINE config 2
// that is 145 GB over the L2 <-> Infinity-Cache fabric (one B block per product: the fabric's ceiling) AND 2 x 4232 bytes
// of ds_write_b128 per product next to the fragment reads -- the LDS pipe is as busy as the matrix pipe.  The XCD-wide tile
// dataflow (mm_tile.h) cut the fabric bytes but not the LDS bytes, and paid for a 256-wave k window.  Here the unit that shares is
// the CU, and what it shares lives in LDS:
//   * a workgroup (8 waves = one CU, two waves per SIMD, persistent) owns a tile of 24 block rows x 3 block columns of C; wave w
//     owns rows 3w .. 3w+2 (a 3 x 3 sub-tile: 81 fp64 accumulators per lane, as in mm_tile.h) -- shape 0; shape 1 (below): 16 waves
//     with 2 x 2 sub-tiles, tiles of 32 x 2;
//   * the B blocks (k, j) of the tile's three columns -- needed by every wave that has an A block in inner block k, 2.4 of the 8
//     waves on average at 10 % fill -- are fetched ONCE per CU into a shared ring of D slots in LDS: 0.38 B blocks per product
//     cross the fabric instead of 1.0, and reach LDS once instead of once per product;
//   * A blocks go to a private two-slot ring per wave (0.9 per product; a product that shares its A block with the previous one
//     skips the copy) and come out of the XCD's L2: the 32 CUs of an XCD work on the SAME 24 block rows at about the same k, so an
//     A block is used by 8.7 CUs while it is L2-resident -- IF the CUs stay together: left alone they drift apart by more than the L2
//     holds (measured: L2 hit rate 0.19, 140 GB over the fabric), so the waves of an XCD keep a k window as in mm_tile.h (every wave
//     publishes the position it will fetch next, nobody fetches beyond the minimum + W); the eight waves of a CU are already held
//     together by the ring, so the spread that costs waiting is that of 32 CUs with 1026 +- 35 products per tile, not of 256 waves;
//   * the ring is a tiny cache with reference counts, one state word per slot in LDS: the first wave that needs B block n (its
//     sequence number in the CU's k-sorted sweep) and finds slot n % D free claims it (compare-and-swap), copies the block by
//     LDS-DMA and publishes it one product later, when its own in-order wait has covered the copy; every user decrements the count
//     after its last product with the block.  A wave runs ahead of the slowest user of the ring by at most D blocks; its SIMD
//     partner takes the matrix pipe meanwhile.  The sweep never stops at tile boundaries: a CU's tiles form ONE sequence, and a
//     wave writes its nine C blocks and goes on while others still finish theirs.
// Per product: 5 x 1.28 DMA pieces instead of 10 staging loads + 10 ds_write_b128; fragment reads as single ds_read_b64.
// Index work (band_* kernels in mm_band_index.h): per (tile, wave) one k-sorted product list whose entries carry the B sequence
// number, the number of users of that B block and the two flags (new A block / last use of the B block); integer work, bit-exact by
// construction, checked against the per-block product counts.
// Results never depend on where workgroups run or on how the waves interleave: only WHEN a block is in the ring does.
#ifndef DBCSR_AMD_MM_BAND_H
#define DBCSR_AMD_MM_BAND_H

#include "dma_lds.h"

namespace dbcsr_amd {

// Two SHAPES of the dataflow (DBCSR_AMD_MM_BAND_SHAPE), both with tiles three or two block columns wide and one sub-tile per wave, stacked
// along the rows:
//   0: 8 waves per workgroup (two per SIMD, up to 256 registers), sub-tiles of 3 x 3 C blocks: tiles of 24 x 3; two A slots per wave (the
//      next product's A block is copied while the current one is multiplied)
//   1: 16 waves per workgroup (FOUR per SIMD, up to 128 registers), sub-tiles of 2 x 2: tiles of 32 x 2, 0.30 B blocks per product; ONE A
//      slot per wave (its copy is requested when the previous product's fragments have been read: three other waves of the SIMD cover the
//      wait).  Why: measured on shape 0 (profiles/r04_band_*), a wave spends 1250 cycles per product outside its 864 cycles of MFMAs (copy
//      issue, ring protocol, list handling), and with two waves per SIMD one partner cannot cover that -- the matrix pipe was 45 % busy
//      with a ring that never made anybody wait in the model.
constexpr int kBandMaxWaves = 16;  // waves per workgroup = sub-tiles per tile, at most
constexpr int kBandMaxT = 3;       // largest sub-tile edge
constexpr int kBandSlots = 9;      // C blocks per sub-tile, at most (slot = tc * ti + tj)

// list entry, 16 bytes
struct BandEntry {
  uint32_t a_lo, b_lo;  // low 32 bits of the element offsets into the A / B data areas
  uint32_t w;           // bits 0-3: accumulator slot (3 ti + tj); 4: the A block differs from the previous entry's (copy it); 5: last product
                        // of this wave with the B block (release it); 6: not a product (end-of-tile marker); 7: end of tile: write the
                        // sub-tile's C blocks after this entry; bits 8-15: low bits of the sweep position (inner block >> kshift);
                        // 16-23 / 24-31: bits 32-39 of the A / B offset
  uint32_t s;           // bits 0-22: sequence number of the B block in the CU's sweep; bits 23-27: waves that use the block; bits 28-31:
                        // high bits of the sweep position
};
constexpr uint32_t kBandNewA = 16u, kBandLastB = 32u, kBandNop = 64u, kBandFlush = 128u;

struct BandRem {  // a product whose inner block has another size than K: added to the finished C block afterwards
  uint32_t a_lo, b_lo;
  uint32_t w;  // bits 0-3 slot, 8-15 k extent, 16-23 / 24-31 high offset bits
  uint32_t pad;
};

struct BandDesc {  // one sub-tile
  int64_t c_off[kBandSlots];    // element offset in C_out data, -1: no C block in this slot
  int64_t cin_off[kBandSlots];  // element offset in C_in data, -1: the block is new
};

struct BandGeom {
  int nfr, nfc;     // block rows / columns of the dominant size
  int waves, tr, tc;  // the shape: sub-tiles (= waves) per tile, C block rows / columns of a sub-tile
  int nBR, nBC;     // tile grid: bands of waves * tr rows x groups of tc columns
  int ntiles;       // nBR * nBC, band-major
  int cu_per_xcd;   // workgroups per XCD (a workgroup b works for XCD b % 8)
  int max_i;        // tiles per workgroup, at most
  int kshift, kspan;  // sweep position of an inner block k: k >> kshift (< 4096); kspan > the largest: position of (tile i, k) = i * kspan + (k >> kshift)
  // XCD x sweeps the tiles [x * ntiles / 8, (x + 1) * ntiles / 8); its workgroup c takes every cu_per_xcd-th of them, starting at c
  __host__ __device__ int64_t lo(int x) const { return (int64_t)x * ntiles / 8; }
};

struct BandArgs {
  const BandDesc* descs;      // [waves * nBR][nBC]: sub-tile (waves * band + w, ct)
  const BandEntry* entries;
  const int64_t* list_off;    // [(8 cu_per_xcd * waves) * max_i + 1]: lists in processing order (workgroup, wave, tile of the workgroup)
  const double* a_data;
  const double* b_data;
  double* c_out;
  const double* c_in;
  double alpha, beta;
  BandGeom G;
  unsigned* prog;             // [8][512] next sweep position each wave of an XCD will fetch (zeroed before the launch)
  int window;                 // a wave fetches operands for position p only while p <= (minimum over its XCD) + window; <= 0: no throttle
  int* flags;                 // [0] spins that gave up (must be 0: results are wrong otherwise), [1] list mismatches (index kernels),
                              // [2] largest sequence number, [3] waves that switched their throttle off (speed only)
  unsigned long long* times;  // knob bit 0: [0] total [1] waits for A [2] waits for B [3] multiplies [4] epilogues (10 ns), [5] waves, [6] claims at the last moment, [7] B waits counted
  int knobs;                  // bit 0: timing; bits 1-3: publish quantum = window >> this (default 3)
};

// block sizes the band kernels are built for (cubes)
#define DBCSR_AMD_BAND_SIZES(X) X(23)

// geometry of a shape; false: no such shape
bool band_shape(int shape, int* waves, int* tr, int* tc);
// LDS bytes of a workgroup for ring depth D (0: no kernel for this size / shape / depth)
int band_lds_bytes(int m, int n, int k, int shape, int depth);
// the persistent kernel (8 * cu_per_xcd workgroups); bpol: cache policy of the B copies (0 default, 1 nt); 0 = launched, 1 = no kernel
int band_launch(int m, int n, int k, int shape, int depth, int bpol, unsigned nwg, hipStream_t st, const BandArgs& P);
int band_launch_remainder(int m, int n, hipStream_t st, int64_t nsub, const BandDesc* descs, const int64_t* rem_start, const BandRem* rem,
                          const double* a_data, const double* b_data, double* c_out, double alpha);

}  // namespace dbcsr_amd
#endif
