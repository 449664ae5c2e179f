/* mm_lab_api.h -- entry points that exist in the LAB build only (libdbcsr_acc_amd_lab.so, -DDBCSR_AMD_EXPERIMENTS; dbcsr_amd/csrc/Makefile).
 * Not part of include/: the shipping library exports nothing of this (tests/test_kernel_resources.py::test_shipping_build_holds_no_experiment). */
#ifndef DBCSR_AMD_MM_LAB_API_H
#define DBCSR_AMD_MM_LAB_API_H
#if defined(__cplusplus)
extern "C" {
#endif

/* Diagnostics of the tile kernel (dbcsr_amd/csrc/mm_tile.h) in the last dbcsr_amd_mm_numeric of this handle: waves that gave up
 * waiting in the team's k window (they went on unthrottled: speed only), sub-tiles whose product list disagreed with the per-block
 * product counts (must be 0).  Returns 1 when that call did not run the tile kernel.  Synchronises the device. */
int dbcsr_amd_mm_tile_stats(void* handle, int* waves_gave_up, int* list_mismatches);

/* The same for the band kernel (dbcsr_amd/csrc/mm_band.h: CU-wide C tiles, B shared in an LDS ring): waits of the ring protocol that
 * gave up (must be 0: a block was used before it had landed), (tile, wave) lists that disagreed with the per-block product counts
 * (must be 0).  Returns 1 when the last dbcsr_amd_mm_numeric of this handle did not run the band kernel.  Synchronises the device. */
int dbcsr_amd_mm_band_stats(void* handle, int* waits_gave_up, int* list_mismatches);

#if defined(__cplusplus)
}
#endif
#endif
