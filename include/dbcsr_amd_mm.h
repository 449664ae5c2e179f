/*
 * dbcsr_amd_mm.h -- device-resident local multiply of DBCSR, C-ABI.
 *
 * Replaces, for one rank and one Cannon tick, the host-driven chain
 *   dbcsr_mm_multrec_multiply -> dbcsr_mm_csr_multiply_low -> flush_stacks ->
 *   dbcsr_mm_sched_process -> dbcsr_mm_accdrv_process -> libsmm_acc_process,
 *   then dbcsr_mm_multrec_finalize / dbcsr_finalize
 *   (/root/reference/src/mm/dbcsr_mm_multrec.F:263-335, dbcsr_mm_csr.F:178-359,
 *    dbcsr_mm_sched.F:266-382, dbcsr_mm_accdrv.F:433-541,
 *    src/work/dbcsr_work_operations.F:749+)
 * by two calls that keep panels, index and result in HBM:
 *   dbcsr_amd_mm_symbolic : the CSR x CSR symbolic product on the GPU
 *                           (C index = sorted BCSR, what dbcsr_finalize emits)
 *   dbcsr_amd_mm_numeric  : C_out = beta*C_in + alpha*A*B, one wavefront per
 *                           C block, all its products summed in MFMA
 *                           accumulators, each C block written once.
 *
 * A matrix is passed as the reference's BCSR index (core/dbcsr_types.F:376-385:
 * row_p / col_i / blk_p + one data area), 0-based, 64-bit block offsets,
 * blocks column-major.  ALL POINTERS ARE DEVICE POINTERS.
 *
 * Return: 0 ok, non-zero error (message on stderr).  Streams use the handle
 * convention of dbcsr_acc.h (pointer to hipStream_t, NULL = null stream).
 */
#ifndef DBCSR_AMD_MM_H
#define DBCSR_AMD_MM_H

#include <stdint.h>

#include "dbcsr_acc_libsmm.h"

#if defined(__cplusplus)
extern "C" {
#endif

typedef struct dbcsr_amd_bcsr {
  int32_t nblkrows, nblkcols;
  const int32_t* row_blk_size; /* [nblkrows] */
  const int32_t* col_blk_size; /* [nblkcols] */
  int32_t* row_p;              /* [nblkrows+1] */
  int32_t* col_i;              /* [nblks], ascending inside a row */
  int64_t* blk_p;              /* [nblks], element offset of each block in data */
  void* data;                  /* fp64 or fp32 elements */
  int64_t nblks;
  uint64_t index_stamp;        /* generation of the index arrays above: 0 = unknown; a caller that owns them may set a value that it
                                  changes whenever it writes, frees or re-allocates one of them (see dbcsr_amd_mm_trust_plan).  Set to
                                  0 by the library in every matrix it hands out. */
} dbcsr_amd_bcsr;

typedef struct dbcsr_amd_mm_counts {
  int64_t c_nblks;   /* blocks of C_out */
  int64_t c_nze;     /* elements of C_out */
  int64_t nproducts; /* block products A(i,k)*B(k,j) */
  int64_t flop;      /* sum 2*m*n*k over products == dbcsr_multiply's flop (dbcsr_mm_csr.F:350) */
} dbcsr_amd_mm_counts;

/* Per-(m, n, k) statistics of a multiply: what dbcsr_mm_sched keeps per stack (src/mm/dbcsr_mm_sched.F:392-461, the
 * "flops m x n x k" table of dbcsr_print_statistics); here every product runs on the accelerator. */
typedef struct dbcsr_amd_mnk_stat {
  int32_t m, n, k, reserved;
  int64_t nproducts; /* block products of this size ("matmuls") */
  int64_t flop;      /* 2*m*n*k*nproducts */
} dbcsr_amd_mnk_stat;

int dbcsr_amd_mm_create(void** handle);
int dbcsr_amd_mm_destroy(void* handle);

/* Symbolic product.  c_in may have nblks == 0.  retain_sparsity: C_out keeps
 * exactly C_in's pattern (dbcsr_mm_csr.F:319).  Writes c_out_row_p
 * [nblkrows+1] (device) and *counts (host; the call synchronises `stream`
 * once to deliver them).  The pattern is kept in the handle for the numeric
 * call that must follow with the same A, B, C_in index arrays. */
int dbcsr_amd_mm_symbolic(void* handle, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in,
  int retain_sparsity, int32_t* c_out_row_p, dbcsr_amd_mm_counts* counts, void* stream);

/* Symbolic product with on-the-fly filtering (dbcsr_mm_csr.F:276, dbcsr_mm_cannon.F:1040-1113): a product
 * A(i,k)*B(k,j) is skipped when ||A(i,k)||^2 * ||alpha*B(k,j)||^2 < (filter_eps / max(1, #blocks of A row i))^2
 * (single precision, as the reference); new C blocks are created only by surviving products.  Needs the data
 * areas of a and b (norms) and the alpha of the numeric call that follows.  filter_eps <= 0: same as above. */
int dbcsr_amd_mm_symbolic_filtered(void* handle, libsmm_acc_data_t datatype, double alpha, double filter_eps, const dbcsr_amd_bcsr* a,
  const dbcsr_amd_bcsr* b, const dbcsr_amd_bcsr* c_in, int retain_sparsity, int32_t* c_out_row_p, dbcsr_amd_mm_counts* counts,
  void* stream);

/* Final block filter of a multiply (dbcsr_mm_multrec.F:694-748) / dbcsr_filter: blocks with sum x^2 < eps^2 are
 * dropped.  _count writes new_row_p [nblkrows+1] (device) and the new block/element counts (host, synchronises);
 * _apply then compacts index and data into caller-allocated dst arrays (dst->row_p = new_row_p). */
int dbcsr_amd_bcsr_filter_count(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, double eps, int32_t* new_row_p,
  int64_t* new_nblks, int64_t* new_nze, void* stream);
int dbcsr_amd_bcsr_filter_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, dbcsr_amd_bcsr* dst, void* stream);

/* Submatrix limits of dbcsr_multiply (first_row ... last_k, src/mm/dbcsr_mm.F:631-709): dbcsr_crop_matrix
 * (src/ops/dbcsr_operations.F:1652-1833) keeps the blocks that intersect the window [row_lo, row_hi] x [col_lo, col_hi]
 * (0-based inclusive ELEMENT indices of the full matrix; a negative bound = no bound) and clears the parts of the
 * boundary blocks outside it.  Same two-step protocol as the filter: _count writes new_row_p and the new counts
 * (synchronises), _apply compacts into caller-allocated dst arrays.  _scale_window is dbcsr_scale with limits: in place,
 * only the elements inside the window are multiplied by beta. */
int dbcsr_amd_bcsr_crop_count(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, int64_t row_lo, int64_t row_hi,
  int64_t col_lo, int64_t col_hi, int32_t* new_row_p, int64_t* new_nblks, int64_t* new_nze, void* stream);
int dbcsr_amd_bcsr_crop_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, dbcsr_amd_bcsr* dst, void* stream);
int dbcsr_amd_bcsr_scale_window(void* handle, libsmm_acc_data_t datatype, dbcsr_amd_bcsr* m, double beta, int64_t row_lo,
  int64_t row_hi, int64_t col_lo, int64_t col_hi, void* stream);

/* Numeric phase.  c_out->row_p is the array written by the symbolic call;
 * col_i [c_nblks], blk_p [c_nblks] and data [c_nze] are allocated by the caller
 * and filled here (blocks laid out in index order).  c_out->data may alias
 * c_in->data only when retain_sparsity was set (same pattern, in place) AND c_in's blocks are already laid out
 * packed in index order (blk_p = running sum of the block sizes) -- which is how every matrix produced by this
 * library is laid out; a C_in with another placement must not be aliased (c_out->blk_p is rewritten to the packed
 * offsets).
 * datatype: dbcsr_type_real_8 or dbcsr_type_real_4.  Asynchronous on stream. */
int dbcsr_amd_mm_numeric(void* handle, libsmm_acc_data_t datatype, double alpha, const dbcsr_amd_bcsr* a, const dbcsr_amd_bcsr* b,
  double beta, const dbcsr_amd_bcsr* c_in, dbcsr_amd_bcsr* c_out, void* stream);

/* Structure-only companion of the symbolic call, for multiplies whose products arrive in
 * several passes (the Cannon ticks of dbcsr_mm_cannon.F:1347-1704): emits C_out's index
 * (col_i, blk_p; row_p came from the symbolic call) and sets C_out = beta*C_in on the
 * blocks C_in has, 0 on the new ones.  The passes then call symbolic(retain_sparsity=1) +
 * numeric with c_out aliasing c_in and beta = 1: blocks that get no product in a pass are
 * not touched. */
int dbcsr_amd_mm_init_c(void* handle, libsmm_acc_data_t datatype, double beta, const dbcsr_amd_bcsr* c_in, dbcsr_amd_bcsr* c_out,
  void* stream);

/* Transposed copy of a BCSR matrix on the device (dbcsr_new_transposed,
 * src/ops/dbcsr_transformations.F): dst index arrays/data are caller-allocated
 * with src's nblks / nze; dst->row_blk_size/col_blk_size must already hold the
 * swapped size arrays. */
int dbcsr_amd_bcsr_transpose(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, dbcsr_amd_bcsr* dst, void* stream);

/* Checksums of dbcsr_checksum (src/dist/dbcsr_dist_util.F:432-577) on the
 * device: out[0] = sum x^2, out[1] = sum x*ln|row*col| (1-based global element
 * coordinates).  Synchronises stream. */
int dbcsr_amd_bcsr_checksum(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, double* out2, void* stream);

/* Synthetic block values of the reference's test generator
 * (src/ops/dbcsr_test_methods.F:423-429 + LAPACK dlarnv/slarnv idist=1): block b of the
 * index gets larnv(seed(row+1, nblkrows, col+1, nblkcols, counter)).  Used by the
 * benchmark to create inputs directly in HBM. */
int dbcsr_amd_bcsr_fill_random(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, int counter, void* stream);

/* Same for one rank's part of a distributed matrix: row_gid/col_gid (device, may be NULL)
 * map local block rows/columns to global ones, nblkrows_global is the global row count
 * that enters the seed (values are a pure function of the GLOBAL block coordinates, as in
 * the reference, so any process grid generates the same matrix). */
int dbcsr_amd_bcsr_fill_random_dist(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* m, int counter,
  const int32_t* row_gid, const int32_t* col_gid, int32_t nblkrows_global, void* stream);

/* dbcsr_multiply for one rank as one call (src/dbcsr_api.F:1411-1433 -> src/mm/dbcsr_mm.F:336 dbcsr_multiply_generic):
 *   C_out = beta*C + alpha*op(A)*op(B)
 * transa/transb 'N' | 'T' | 'C' (real data: 'C' == 'T'); limits = {first_row, last_row, first_column, last_column, first_k,
 * last_k}, 1-based inclusive full-matrix indices, 0 = not given, NULL = no limits (inside the window beta scales C, outside it C
 * is unchanged); retain_sparsity and filter_eps as in the reference.  c_out: row_p / col_i / blk_p / data are allocated by the
 * library (its caching device allocator) and belong to the caller afterwards -- ONLY dbcsr_amd_bcsr_release frees them; the size arrays are borrowed
 * from matrix_c.  *flop (may be NULL) receives the reference's flop count.  Returns when the result is complete. */
int dbcsr_amd_multiply(void* handle, char transa, char transb, libsmm_acc_data_t datatype, double alpha, const dbcsr_amd_bcsr* matrix_a,
  const dbcsr_amd_bcsr* matrix_b, double beta, const dbcsr_amd_bcsr* matrix_c, const int64_t* limits, int retain_sparsity,
  double filter_eps, dbcsr_amd_bcsr* c_out, int64_t* flop, void* stream);
int dbcsr_amd_bcsr_release(dbcsr_amd_bcsr* m);

/* HIP-event timing of the last dbcsr_amd_mm_numeric call on this handle, taken
 * on the stream the kernels were launched on: ms_fill = product-list/index
 * emission kernel, ms_numeric = the block-GEMM kernel.  Waits for that call to
 * finish.  Used by bench.py for the roofline figure. */
int dbcsr_amd_mm_timing(void* handle, float* ms_fill, float* ms_numeric);

/* Symbol name of the dominant kernel, for profile look-up. */
/* Symmetric operands (src/core/dbcsr_types.F: matrix_type 'S' / 'A'; the reference desymmetrizes them while it builds the
 * multiplication images, src/mm/dbcsr_mm_cannon.F:284, 351-379): src holds ONE block per symmetric pair (any mix of upper and
 * lower blocks, square block structure); the full matrix gets block (c, r) = +block(r, c)^T (symmetric) or -block(r, c)^T
 * (antisymmetric) in addition.  _count writes dst_row_p [nblkrows+1] (device) and the block / element counts (host,
 * synchronises); _apply fills caller-allocated dst arrays (dst->row_p = that row_p), blocks packed in index order. */
int dbcsr_amd_bcsr_desymmetrize_count(void* handle, const dbcsr_amd_bcsr* src, int32_t* dst_row_p, int64_t* nblks, int64_t* nze, void* stream);
int dbcsr_amd_bcsr_desymmetrize_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, int antisymmetric,
  dbcsr_amd_bcsr* dst, void* stream);

/* Product matrix WITH symmetry (matrix_type 'S' / 'A' of matrix_c).  The reference puts the index of such a product matrix into
 * canonical (checkerboard) form before the multiplication (src/mm/dbcsr_mm.F:711-719, dbcsr_make_index_canonical), its local
 * multiply computes block (i, j) only when that is the stored one of the pair (i, j) / (j, i) (src/mm/dbcsr_mm_csr.F:280-292,
 * checker_tr of src/dist/dbcsr_dist_operations.F:65-75), and the result is returned as the stored triangle.  The pieces:
 *   dbcsr_amd_bcsr_twin_{count,apply}  mode 0: desymmetrize (= the two calls above); mode 1: stored triangle (row <= column) ->
 *     canonical form; mode 2: canonical form -> stored triangle.  A block that changes sides is transposed (negated when
 *     antisymmetric).  Same calling convention as desymmetrize_{count,apply}.
 *   dbcsr_amd_mm_set_canonical_product(handle, 1): the following symbolic phases of this handle leave out the products of
 *     blocks that are not stored in canonical form (blocks of C_in are kept wherever they are); 0 switches it off again.
 *   dbcsr_amd_multiply_symmetric_c: the whole sequence in one call; matrix_c and c_out hold the stored triangle (row <= column),
 *     no limits (the reference's own tests run symmetric products with full limits only, tests/dbcsr_test_multiply.F:196-200). */
/* desymmetrize in one call: dst's arrays are allocated by the library (dbcsr_amd_bcsr_release frees them), size arrays borrowed from src */
int dbcsr_amd_bcsr_desymmetrized(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, int antisymmetric, dbcsr_amd_bcsr* dst,
  void* stream);
int dbcsr_amd_bcsr_twin_count(void* handle, const dbcsr_amd_bcsr* src, int mode, int32_t* dst_row_p, int64_t* nblks, int64_t* nze, void* stream);
int dbcsr_amd_bcsr_twin_apply(void* handle, libsmm_acc_data_t datatype, const dbcsr_amd_bcsr* src, int mode, int antisymmetric,
  dbcsr_amd_bcsr* dst, void* stream);
int dbcsr_amd_mm_set_canonical_product(void* handle, int on);
int dbcsr_amd_multiply_symmetric_c(void* handle, char transa, char transb, libsmm_acc_data_t datatype, double alpha,
  const dbcsr_amd_bcsr* matrix_a, const dbcsr_amd_bcsr* matrix_b, double beta, const dbcsr_amd_bcsr* matrix_c, int antisymmetric,
  int retain_sparsity, double filter_eps, dbcsr_amd_bcsr* c_out, int64_t* flop, void* stream);
/* the same with limits on the inner dimension (dbcsr_multiply's first_k / last_k: 1-based inclusive element indices, 0 = not given) -- what the
   reference's own tests of products with symmetry use (tests/dbcsr_test_multiply.F:196-200: full row / column limits, any k limits) */
int dbcsr_amd_multiply_symmetric_c_klimits(void* handle, char transa, char transb, libsmm_acc_data_t datatype, double alpha,
  const dbcsr_amd_bcsr* matrix_a, const dbcsr_amd_bcsr* matrix_b, double beta, const dbcsr_amd_bcsr* matrix_c, int antisymmetric,
  int64_t first_k, int64_t last_k, int retain_sparsity, double filter_eps, dbcsr_amd_bcsr* c_out, int64_t* flop, void* stream);

/* Statistics of the last dbcsr_amd_mm_numeric of this handle, by (m, n, k): at most max_entries records are written to
 * `out` (host memory), *n_entries receives the number of distinct triples (larger than max_entries = truncated).
 * Counted on the device from the product lists of that call; synchronises `stream`. */
int dbcsr_amd_mm_stats(void* handle, dbcsr_amd_mnk_stat* out, int max_entries, int* n_entries, void* stream);

const char* dbcsr_amd_mm_kernel_name(libsmm_acc_data_t datatype);
/* name (with its template arguments) of the block-product kernel the last dbcsr_amd_mm_numeric of this handle launched */
const char* dbcsr_amd_mm_last_kernel(void* handle);
/* name of the stack kernel the calling thread's last libsmm_acc_process (fp64, homogeneous stack) launched: "smm_stack_f64_exact<m,n,k>" (compiled
   for the triplet at run time, as libsmm_acc.cpp:90-195 does), "smm_stack_f64_lds(...)" (run-time sizes), "smm_stack_f64_big(...)" (blocks of 33 ... 80) */
const char* dbcsr_amd_smm_last_kernel(void);

/* Plan reuse: a multiply whose A, B and C_in have exactly the index arrays (row_p, col_i, blk_p, block sizes) of the previous
 * multiply of this handle -- every SCF step of a CP2K run -- skips its symbolic phase: the engine keeps device copies of the last
 * call's index arrays and compares the incoming ones on the device (one small kernel, one flag).  Multiplies with filter_eps > 0
 * never reuse (their pattern depends on the values).  DBCSR_AMD_MM_PLAN=0 switches it off.  Counters since the handle was made: */
/* Plan reuse without the comparison: while `on`, operands whose twelve index arrays (row_p / col_i / blk_p of A, B, C_in, the three
   block-size arrays) sit at the ADDRESSES the saved plan saw AND carry the non-zero index_stamp values the plan was saved with are
   taken as unchanged -- no comparison kernel, no synchronisation of the stream in dbcsr_amd_mm_symbolic.  The stamp is what makes the
   address test safe: an operand that was freed and allocated again at the same addresses with another pattern has another stamp (or
   none: 0 is never trusted) and is compared on the device as always.  For callers that own these arrays (the panels of a distributed
   multiply, dbcsr_amd/cannon.py; the benchmark's operands). */
int dbcsr_amd_mm_trust_plan(void* handle, int on);

/* A filtered multiply (reference: src/mm/dbcsr_mm_multrec.F:373-383 -- the product of a multiply with filter_eps is filtered with the same eps before it is
   finalized): announce the final block filter of the NEXT dbcsr_amd_mm_numeric of this handle.  Its product kernels form a block's squared norm before they write it
   and leave a block with ||blk||^2 < eps^2 UNWRITTEN -- the block filter is going to drop it (same double, same comparison), nobody may read it before.  The C that
   comes back is therefore only good for dbcsr_amd_bcsr_filter_count / _apply with an eps that is not smaller (a smaller one is refused: -3).  On products with many
   dropped blocks the dropped share of C's write traffic is saved.  Without this call every block is written.  Call it AFTER the symbolic phase of the
   multiply it is meant for (a symbolic phase cancels an announcement that was never consumed).  fp64; ignored for retain_sparsity and in-place accumulation. */
int dbcsr_amd_mm_expect_filter(void* handle, double eps);
int dbcsr_amd_mm_plan_stats(void* handle, int64_t* reused, int64_t* built);

/* Measurement helper (bench.py, roofline.fabric): what the L2 <-> Infinity-Cache fabric of the current device delivers, in TB/s -- a
 * plain streaming read of a 160 MB window by all CUs, and the block gather of the block-product dataflow (4232-byte blocks from
 * pseudo-random places of the window into LDS, whole 128-byte lines counted).  Takes well under a second; synchronises the device. */
int dbcsr_amd_fabric_probe(double* stream_tb_per_s, double* gather_tb_per_s);

#if defined(__cplusplus)
}
#endif
#endif
