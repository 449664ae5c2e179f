"""``dbcsr_multiply`` -- host-side mirror of the reference's operator for the
hot path (src/dbcsr_api.F:1411-1433 -> src/mm/dbcsr_mm.F:336 dbcsr_multiply_generic):

    C <- beta*C + alpha*op(A)*op(B)

Same argument names, meaning and error behaviour; the work is done by the
C-ABI engine (include/dbcsr_amd_mm.h) on the GPU.  One rank / one device here;
the multi-GPU Cannon driver lives in dbcsr_amd/cannon.py."""
import ctypes as C
import math
import os

import torch

from . import lib as _lib
from .matrix import DbcsrMatrix, StreamHandle

dbcsr_no_transpose = "N"
dbcsr_transpose = "T"
dbcsr_conjugate_transpose = "C"


class MultiplyEngine:
    """Owns the native workspace (bitmaps, product lists) across calls."""

    def __init__(self, lab=None):
        # lab: the build with the experimental dataflows (default: DBCSR_AMD_LAB=1 in the environment); both builds may live in one process
        self.lab = _lib.want_lab() if lab is None else bool(lab)
        self.L = _lib.load_library(self.lab)
        self.h = C.c_void_p()
        rc = self.L.dbcsr_amd_mm_create(C.byref(self.h))
        if rc != 0:
            raise RuntimeError("dbcsr_amd_mm_create failed (%d)" % rc)

    def close(self):
        if self.h:
            self.L.dbcsr_amd_mm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def transposed(self, M, stream=None):
        st = StreamHandle(stream)
        out = DbcsrMatrix(M.col_blk_size, M.row_blk_size, torch.empty(M.nblkcols + 1, dtype=torch.int32, device=M.data.device),
                          torch.empty_like(M.col_i), torch.empty_like(M.blk_p), torch.empty_like(M.data), M.name + "^T")
        src, dst = M.desc(), out.desc(out=True)
        rc = self.L.dbcsr_amd_bcsr_transpose(self.h, M.dtype_code, C.byref(src), C.byref(dst), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_transpose failed (%d)" % rc)
        return out

    def checksum(self, M, stream=None):
        st = StreamHandle(stream)
        out = (C.c_double * 2)()
        d = M.desc()
        rc = self.L.dbcsr_amd_bcsr_checksum(self.h, M.dtype_code, C.byref(d), out, st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_checksum failed (%d)" % rc)
        return out[0], out[1]

    def fill_random(self, M, counter, stream=None):
        st = StreamHandle(stream)
        d = M.desc()
        rc = self.L.dbcsr_amd_bcsr_fill_random(self.h, M.dtype_code, C.byref(d), int(counter), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_fill_random failed (%d)" % rc)

    # -- statistics (dbcsr_mm_sched.F:392-461 / dbcsr_print_statistics) -------------------------------------------
    def mnk_statistics(self, stream=None, max_entries=4096):
        """[(m, n, k, nproducts, flop)] of the last numeric call, largest flop first (dbcsr_amd_mm_stats)."""
        st = StreamHandle(stream)
        buf = (_lib.MnkStat * max_entries)()
        n = C.c_int32(0)
        rc = self.L.dbcsr_amd_mm_stats(self.h, buf, max_entries, C.byref(n), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_mm_stats failed (%d)" % rc)
        return [(r.m, r.n, r.k, r.nproducts, r.flop) for r in buf[:min(n.value, max_entries)]]

    def accumulate_statistics(self, stream=None):
        """Adds the last multiply to this engine's running totals (the reference accumulates until dbcsr_print_statistics)."""
        tot = self.__dict__.setdefault("stats_total", {})
        for m, n, k, cnt, fl in self.mnk_statistics(stream):
            c0, f0 = tot.get((m, n, k), (0, 0))
            tot[(m, n, k)] = (c0 + cnt, f0 + fl)
        self.__dict__["stats_multiplications"] = self.__dict__.get("stats_multiplications", 0) + 1
        return tot

    def print_statistics(self, file=None):
        """The 'flops m x n x k' table of dbcsr_print_statistics (src/mm/dbcsr_mm_sched.F:463-560); every product ran on the ACC."""
        import sys as _sys
        f = file or _sys.stdout
        tot = self.__dict__.get("stats_total", {})
        total = sum(v[1] for v in tot.values())
        print(" " + "-" * 79, file=f)
        print(" -%s-" % "DBCSR STATISTICS".center(77), file=f)
        print(" " + "-" * 79, file=f)
        print(" COUNTER                                    TOTAL       BLAS       SMM       ACC", file=f)
        for (m, n, k), (cnt, fl) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
            print(" flops %5d x %5d x %5d %20d       0.0%%      0.0%%    100.0%%" % (m, n, k, fl), file=f)
        print(" flops total %30.6E       0.0%%      0.0%%    100.0%%" % float(total), file=f)
        print(" matmuls total %28d       0.0%%      0.0%%    100.0%%" % sum(v[0] for v in tot.values()), file=f)
        print(" # multiplications %24d" % self.__dict__.get("stats_multiplications", 0), file=f)
        print(" " + "-" * 79, file=f)

    def last_kernel(self):
        """name of the block-product kernel the last numeric call launched (dbcsr_amd_mm_last_kernel; with k passes: the last pass's)"""
        eng = getattr(self, "_last_pass_engine", None) if getattr(self, "last_kchunks", 1) > 1 else None
        v = self.L.dbcsr_amd_mm_last_kernel((eng or self).h)
        return v.decode() if v else ""

    def plan_stats(self):
        """(multiplies that reused the previous plan, multiplies that ran their symbolic phase) since the engine was made"""
        a, b = C.c_int64(), C.c_int64()
        if self.L.dbcsr_amd_mm_plan_stats(self.h, C.byref(a), C.byref(b)) != 0:
            raise RuntimeError("dbcsr_amd_mm_plan_stats failed")
        return a.value, b.value

    def trust_plan(self, on=True):
        """Plan reuse by address (dbcsr_amd_mm_trust_plan): for operands whose index arrays are never written in place."""
        if self.L.dbcsr_amd_mm_trust_plan(self.h, 1 if on else 0) != 0:
            raise RuntimeError("dbcsr_amd_mm_trust_plan failed")

    def tile_stats(self):
        """(waves that gave up waiting in the k window, sub-tile lists that disagreed with the per-block counts) of the last numeric
        call, or None when it did not run the tile kernel (dbcsr_amd_mm_tile_stats)"""
        a, b = C.c_int(), C.c_int()
        rc = self.L.dbcsr_amd_mm_tile_stats(self.h, C.byref(a), C.byref(b))
        if rc < 0:
            raise RuntimeError("dbcsr_amd_mm_tile_stats failed (%d)" % rc)
        return None if rc else (a.value, b.value)

    def band_stats(self):
        """(waits of the ring protocol that gave up, (tile, wave) lists that disagreed with the per-block counts) of the last numeric
        call -- both must be 0 --, or None when it did not run the band kernel (dbcsr_amd_mm_band_stats)"""
        a, b = C.c_int(), C.c_int()
        rc = self.L.dbcsr_amd_mm_band_stats(self.h, C.byref(a), C.byref(b))
        if rc < 0:
            raise RuntimeError("dbcsr_amd_mm_band_stats failed (%d)" % rc)
        return None if rc else (a.value, b.value)

    def _last_timing(self):
        f, n = C.c_float(), C.c_float()
        rc = self.L.dbcsr_amd_mm_timing(self.h, C.byref(f), C.byref(n))
        if rc != 0:
            raise RuntimeError("dbcsr_amd_mm_timing failed (%d)" % rc)
        return f.value, n.value

    def fill_random_dist(self, M, counter, row_gid, col_gid, nblkrows_global, stream=None):
        st = StreamHandle(stream)
        d = M.desc()
        rc = self.L.dbcsr_amd_bcsr_fill_random_dist(self.h, M.dtype_code, C.byref(d), int(counter), row_gid.data_ptr(),
                                                    col_gid.data_ptr(), int(nblkrows_global), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_fill_random_dist failed (%d)" % rc)

    # -- the phases of a multiply, separately (used by the Cannon driver) -------------
    def symbolic(self, A, B, Cm, retain_sparsity=False, stream=None):
        """Returns (row_p of C_out, counts)."""
        st = StreamHandle(stream)
        a, b, cin = A.desc(), B.desc(), Cm.desc()
        row_p = torch.empty(Cm.nblkrows + 1, dtype=torch.int32, device=Cm.row_p.device)
        counts = _lib.MmCounts()
        rc = self.L.dbcsr_amd_mm_symbolic(self.h, C.byref(a), C.byref(b), C.byref(cin), 1 if retain_sparsity else 0,
                                          row_p.data_ptr(), C.byref(counts), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_mm_symbolic failed (%d)" % rc)
        return row_p, counts

    def init_c(self, beta, Cm, row_p, counts, dtype, stream=None):
        """C_out with the pattern of the last symbolic call, = beta*Cm on Cm's blocks, 0 elsewhere."""
        st = StreamHandle(stream)
        dev = Cm.row_p.device
        out = DbcsrMatrix(Cm.row_blk_size, Cm.col_blk_size, row_p, torch.empty(counts.c_nblks, dtype=torch.int32, device=dev),
                          torch.empty(counts.c_nblks, dtype=torch.int64, device=dev),
                          torch.empty(counts.c_nze, dtype=dtype, device=dev), Cm.name)
        cin, cout = Cm.desc(), out.desc(out=True)
        rc = self.L.dbcsr_amd_mm_init_c(self.h, _lib.dbcsr_type_real_8 if dtype == torch.float64 else _lib.dbcsr_type_real_4,
                                        float(beta), C.byref(cin), C.byref(cout), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_mm_init_c failed (%d)" % rc)
        return out

    def accumulate(self, alpha, A, B, Cacc, stream=None):
        """Cacc += alpha*A*B restricted to Cacc's pattern, in place; returns counts of this pass."""
        st = StreamHandle(stream)
        a, b, c = A.desc(), B.desc(), Cacc.desc()
        row_p = torch.empty(Cacc.nblkrows + 1, dtype=torch.int32, device=Cacc.row_p.device)
        counts = _lib.MmCounts()
        rc = self.L.dbcsr_amd_mm_symbolic(self.h, C.byref(a), C.byref(b), C.byref(c), 1, row_p.data_ptr(), C.byref(counts), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_mm_symbolic failed (%d)" % rc)
        cout = _lib.BcsrDesc(c.nblkrows, c.nblkcols, c.row_blk_size, c.col_blk_size, row_p.data_ptr(), c.col_i, c.blk_p, c.data, c.nblks)
        rc = self.L.dbcsr_amd_mm_numeric(self.h, Cacc.dtype_code, float(alpha), C.byref(a), C.byref(b), 1.0, C.byref(c), C.byref(cout),
                                         st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_mm_numeric failed (%d)" % rc)
        return counts

    accepts_out_data = True

    def numeric_after_symbolic(self, alpha, A, B, beta, Cm, row_p, counts, dtype, stream=None, out_data=None):
        """Second half of multiply_local for a symbolic() call made earlier on the same operands.  out_data: where the blocks of C_out
        go (a tensor of counts.c_nze elements, e.g. a slice of a larger buffer) instead of a new allocation."""
        st = StreamHandle(stream)
        dev = Cm.row_p.device
        if out_data is not None and (out_data.numel() != counts.c_nze or out_data.dtype != dtype or not out_data.is_contiguous()):
            raise ValueError("numeric_after_symbolic: out_data must be a contiguous tensor of c_nze = %d elements of %s" % (counts.c_nze, dtype))
        out = DbcsrMatrix(Cm.row_blk_size, Cm.col_blk_size, row_p, torch.empty(counts.c_nblks, dtype=torch.int32, device=dev),
                          torch.empty(counts.c_nblks, dtype=torch.int64, device=dev),
                          out_data if out_data is not None else torch.empty(counts.c_nze, dtype=dtype, device=dev), Cm.name)
        a, b, cin, cout = A.desc(), B.desc(), Cm.desc(), out.desc(out=True)
        rc = self.L.dbcsr_amd_mm_numeric(self.h, out.dtype_code, float(alpha), C.byref(a), C.byref(b), float(beta), C.byref(cin),
                                         C.byref(cout), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_mm_numeric failed (%d)" % rc)
        return out

    def filtered(self, M, eps, stream=None):
        """Copy of M without the blocks whose squared Frobenius norm is below eps^2 (final filter of a multiply)."""
        st = StreamHandle(stream)
        dev = M.row_p.device
        src = M.desc()
        row_p = torch.empty(M.nblkrows + 1, dtype=torch.int32, device=dev)
        nb, nz = C.c_int64(), C.c_int64()
        rc = self.L.dbcsr_amd_bcsr_filter_count(self.h, M.dtype_code, C.byref(src), float(eps), row_p.data_ptr(), C.byref(nb), C.byref(nz),
                                                st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_filter_count failed (%d)" % rc)
        if nb.value == M.nblks:   # nothing falls below the threshold: no second copy of the matrix
            return M
        out = DbcsrMatrix(M.row_blk_size, M.col_blk_size, row_p, torch.empty(nb.value, dtype=torch.int32, device=dev),
                          torch.empty(nb.value, dtype=torch.int64, device=dev), torch.empty(nz.value, dtype=M.dtype, device=dev), M.name)
        dst = out.desc(out=True)
        rc = self.L.dbcsr_amd_bcsr_filter_apply(self.h, M.dtype_code, C.byref(src), C.byref(dst), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_filter_apply failed (%d)" % rc)
        return out

    def desymmetrized(self, M, stream=None):
        """Full matrix of a symmetric ('S') / antisymmetric ('A') operand (dbcsr_desymmetrize_deep, done by the reference while
        it builds the multiplication images, dbcsr_mm_cannon.F:284, 351-379)."""
        if M.symmetry == "N":
            return M
        if M.symmetry not in ("S", "A"):
            raise ValueError("unsupported matrix symmetry %r (real data: 'N', 'S', 'A')" % (M.symmetry,))
        st = StreamHandle(stream)
        src = M.desc()
        row_p = torch.empty(M.nblkrows + 1, dtype=torch.int32, device=M.row_p.device)
        nb, nz = C.c_int64(0), C.c_int64(0)
        rc = self.L.dbcsr_amd_bcsr_desymmetrize_count(self.h, C.byref(src), row_p.data_ptr(), C.byref(nb), C.byref(nz), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_desymmetrize_count failed (%d)" % rc)
        dev = M.row_p.device
        out = DbcsrMatrix(M.row_blk_size, M.col_blk_size, row_p, torch.empty(nb.value, dtype=torch.int32, device=dev),
                          torch.empty(nb.value, dtype=torch.int64, device=dev), torch.empty(nz.value, dtype=M.dtype, device=dev), M.name)
        d = out.desc(out=True)
        rc = self.L.dbcsr_amd_bcsr_desymmetrize_apply(self.h, M.dtype_code, C.byref(src), 1 if M.symmetry == "A" else 0, C.byref(d), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_desymmetrize_apply failed (%d)" % rc)
        return out

    def twin_moved(self, M, mode, symmetry, stream=None):
        """Blocks of a matrix with symmetry moved to their twins (r, c) -> (c, r), transposed (negated when antisymmetric):
        mode 1 = stored triangle -> canonical (checkerboard) form (dbcsr_make_index_canonical), mode 2 = canonical form -> stored
        triangle (row <= column).  include/dbcsr_amd_mm.h: dbcsr_amd_bcsr_twin_count / _apply."""
        st = StreamHandle(stream)
        src = M.desc()
        dev = M.row_p.device
        row_p = torch.empty(M.nblkrows + 1, dtype=torch.int32, device=dev)
        nb, nz = C.c_int64(0), C.c_int64(0)
        rc = self.L.dbcsr_amd_bcsr_twin_count(self.h, C.byref(src), mode, row_p.data_ptr(), C.byref(nb), C.byref(nz), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_twin_count failed (%d)" % rc)
        out = DbcsrMatrix(M.row_blk_size, M.col_blk_size, row_p, torch.empty(nb.value, dtype=torch.int32, device=dev),
                          torch.empty(nb.value, dtype=torch.int64, device=dev), torch.empty(nz.value, dtype=M.dtype, device=dev), M.name)
        d = out.desc(out=True)
        rc = self.L.dbcsr_amd_bcsr_twin_apply(self.h, M.dtype_code, C.byref(src), mode, 1 if symmetry == "A" else 0, C.byref(d), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_twin_apply failed (%d)" % rc)
        return out

    def set_canonical_product(self, on):
        if self.L.dbcsr_amd_mm_set_canonical_product(self.h, 1 if on else 0) != 0:
            raise RuntimeError("dbcsr_amd_mm_set_canonical_product failed")

    @staticmethod
    def empty_like(M):
        """C without any block, same block sizes (the product matrix after the reference has discarded it, dbcsr_mm.F:865-870)."""
        return DbcsrMatrix.empty_like_pattern(M.row_blk_size, M.col_blk_size, M.dtype, device=M.data.device, name=M.name)

    def cropped(self, M, row_bounds=None, col_bounds=None, stream=None):
        """dbcsr_crop_matrix (src/ops/dbcsr_operations.F:1652-1833): copy of M restricted to the window given as 0-based
        inclusive element bounds (None = no bound); boundary blocks keep their size, their outside part is zero."""
        st = StreamHandle(stream)
        dev = M.row_p.device
        src = M.desc()
        r0, r1 = (-1, -1) if row_bounds is None else (int(row_bounds[0]), int(row_bounds[1]))
        c0, c1 = (-1, -1) if col_bounds is None else (int(col_bounds[0]), int(col_bounds[1]))
        row_p = torch.empty(M.nblkrows + 1, dtype=torch.int32, device=dev)
        nb, nz = C.c_int64(), C.c_int64()
        rc = self.L.dbcsr_amd_bcsr_crop_count(self.h, M.dtype_code, C.byref(src), r0, r1, c0, c1, row_p.data_ptr(), C.byref(nb),
                                              C.byref(nz), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_crop_count failed (%d)" % rc)
        out = DbcsrMatrix(M.row_blk_size, M.col_blk_size, row_p, torch.empty(nb.value, dtype=torch.int32, device=dev),
                          torch.empty(nb.value, dtype=torch.int64, device=dev), torch.empty(nz.value, dtype=M.dtype, device=dev), M.name)
        dst = out.desc(out=True)
        rc = self.L.dbcsr_amd_bcsr_crop_apply(self.h, M.dtype_code, C.byref(src), C.byref(dst), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_crop_apply failed (%d)" % rc)
        return out

    def scaled_window(self, M, beta, row_bounds=None, col_bounds=None, stream=None):
        """dbcsr_scale(matrix, beta, limits) on a copy of M: only the elements inside the window are scaled."""
        st = StreamHandle(stream)
        out = DbcsrMatrix(M.row_blk_size, M.col_blk_size, M.row_p, M.col_i, M.blk_p, M.data.clone(), M.name)
        r0, r1 = (-1, -1) if row_bounds is None else (int(row_bounds[0]), int(row_bounds[1]))
        c0, c1 = (-1, -1) if col_bounds is None else (int(col_bounds[0]), int(col_bounds[1]))
        d = out.desc(out=True)
        rc = self.L.dbcsr_amd_bcsr_scale_window(self.h, M.dtype_code, C.byref(d), float(beta), r0, r1, c0, c1, st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_bcsr_scale_window failed (%d)" % rc)
        return out

    # L2 blocking over k: when a block row of A is larger than about a third of an XCD's 4 MB L2 (config 5: 819 blocks of
    # 4 KB = 3.3 MB) neither operand stays cache-resident and every block product pulls both of its blocks through the
    # fabric.  The product is then formed in passes over k ranges, C accumulating in place of the previous pass: per pass
    # the A row and the B panel are 1/n of their size (measured on config 5, one GPU: 2761 ms in one pass, 2147 ms in 4
    # passes, 2357 ms in 8, tools/kchunk_probe.py).  The price -- C re-read and re-written per pass -- is why this is not
    # done for small A rows.
    # (round 5, gpurun_out/r05_s03/sweeps.jsonl: 32768^2 of 23 x 23 at 20 % fill, A rows of 1.2 MB: 83.0 ms in one pass, 77.2 in two)
    # (round 6, session r06_49: after the exact-size kernels' padded LDS pitches a single pass over 32 x 32 blocks is fast, and passes only pay when a C block
    #  collects many products -- the price of a pass is C's bytes, its gain the operands': 32^3 at 10 % fill (14 products per C block, 1.17 MB rows) 44.0 ms in one
    #  pass against 53.0 in two; 32^3 at 20 % (41) 63.6 against 66.3; 28^3 at 15 % (32) 80.6 against 84.3; 23^3 at 20 % (57) 83.5 against 75.3; config 5 (164): four
    #  passes.  Hence the second condition: at least KCHUNK_MIN_PRODUCTS products per C block, estimated from the operands' fills.)
    KCHUNK_ROW_BYTES = 1.0 * 2 ** 20
    KCHUNK_MIN_PRODUCTS = 48.0

    def _auto_kchunks(self, A, filter_eps, B=None):
        if filter_eps and filter_eps > 0:  # the on-the-fly filter counts the blocks of a whole A row (dbcsr_mm_cannon.F:1100-1110)
            return 1
        forced = os.environ.get("DBCSR_AMD_MM_KCHUNKS")
        if forced:
            return max(1, int(forced))
        row_bytes = A.data.numel() * A.data.element_size() / max(1, A.nblkrows)
        if row_bytes <= self.KCHUNK_ROW_BYTES or A.nblkcols < 64:
            return 1
        fill_a = A.nblks / max(1.0, float(A.nblkrows) * A.nblkcols)
        fill_b = fill_a if B is None else B.nblks / max(1.0, float(B.nblkrows) * B.nblkcols)
        if A.nblkcols * fill_a * fill_b < self.KCHUNK_MIN_PRODUCTS:   # (expected products per C block)
            return 1
        if A.data.numel() > 1024 * max(1, A.nblks):
            # blocks above 32 x 32 on average: the workgroup-per-C-block kernel (mm_numeric_f64_big.h) shares its operand slabs through LDS and
            # re-reading and re-writing C per pass costs more than the passes save (72^3 at 30 % fill, 16384^2: 21.5 ms in one pass against
            # 25.5 ms in three, gpurun_out/r05_s02/large_blocks_after.jsonl)
            return 1
        return int(min(8, math.ceil(row_bytes / self.KCHUNK_ROW_BYTES)))   # (pieces of at most the threshold: 1 MB)

    def _kpass_views(self, A, B, n, counts):
        """[(A's block columns [k0, k1), B's block rows [k0, k1), an engine of its own)] for n ranges of inner blocks of about equal
        element count: index arrays over the operands' own data areas.  An engine per pass keeps that pass's product lists between
        multiplies (12 bytes per block product: 33 GB for config 5); when the device is short of that, the passes share ONE engine,
        whose plan is then rebuilt pass by pass as in round 3."""
        need = 12 * int(counts.nproducts) + 200 * int(counts.c_nblks) * n
        free, _total = torch.cuda.mem_get_info(A.data.device)
        shared = None
        if need > 0.4 * free:
            shared = type(self)(lab=self.lab)
        off = torch.cumsum(A.col_blk_size.to(torch.int64), 0).cpu()
        total = int(off[-1])
        bounds = [0]
        for i in range(1, n):   # block boundary nearest to i / n of the elements
            bounds.append(int(torch.searchsorted(off, torch.tensor(round(i * total / n), dtype=torch.int64))) + 1)
        bounds.append(A.nblkcols)
        bounds = sorted(set(min(max(b, 0), A.nblkcols) for b in bounds))
        nbr = A.nblkrows
        rows = torch.repeat_interleave(torch.arange(nbr, device=A.row_p.device), torch.diff(A.row_p.to(torch.int64)))
        views = []
        for k0, k1 in zip(bounds[:-1], bounds[1:]):
            if k1 <= k0:
                continue
            keep = (A.col_i >= k0) & (A.col_i < k1)
            cnt = torch.bincount(rows[keep], minlength=nbr)
            row_p = torch.zeros(nbr + 1, dtype=torch.int32, device=A.row_p.device)
            row_p[1:] = torch.cumsum(cnt, 0).to(torch.int32)
            Ac = DbcsrMatrix(A.row_blk_size, A.col_blk_size[k0:k1].contiguous(), row_p, (A.col_i[keep] - k0).to(torch.int32).contiguous(),
                             A.blk_p[keep].contiguous(), A.data, A.name)
            lo, hi = int(B.row_p[k0]), int(B.row_p[k1])
            Bc = DbcsrMatrix(B.row_blk_size[k0:k1].contiguous(), B.col_blk_size, (B.row_p[k0:k1 + 1] - lo).to(torch.int32).contiguous(),
                             B.col_i[lo:hi].contiguous(), B.blk_p[lo:hi].contiguous(), B.data, B.name)
            e = shared
            if e is None:
                e = type(self)(lab=self.lab)
                e.trust_plan(True)   # the views are this object's own and never written again
            views.append((Ac, Bc, e))
        return views

    def last_timing(self):
        """(ms_fill, ms_numeric) of the last numeric call, from HIP events on its stream (with k passes: of the last pass)."""
        eng = getattr(self, "_last_pass_engine", None) if getattr(self, "last_kchunks", 1) > 1 else None
        return MultiplyEngine._last_timing(eng or self)

    def multiply_local(self, alpha, A, B, beta, Cm, retain_sparsity=False, stream=None, filter_eps=0.0, kchunks=None):
        """C_out = beta*Cm + alpha*A*B for already-oriented operands; returns (C_out, counts)."""
        n = self._auto_kchunks(A, filter_eps, B) if kchunks is None else int(kchunks)
        if filter_eps and filter_eps > 0.0 and n > 1:
            raise ValueError("multiply_local: k passes cannot be combined with filter_eps (the on-the-fly filter counts the blocks of a whole A row)")
        if n > 1 and A.nblkcols >= n:
            # structure once (symbolic product of the whole operands, C = beta*Cm on it), then one in-place pass per k range
            row_p, total = self.symbolic(A, B, Cm, retain_sparsity=retain_sparsity, stream=stream)
            out = self.init_c(beta, Cm, row_p, total, A.dtype, stream=stream)
            # The passes' operands are VIEWS: the block columns [k0, k1) of A and the block rows [k0, k1) of B as index arrays of their own
            # over the operands' own data areas (a block's offset does not care which index names it) -- no copy of 2 x 13.7 GB per
            # multiply at config 5 --, built once per operand pattern and kept with one engine per pass, whose plan (symbolic product,
            # product lists, launch order of THAT pass) is then reused by every later multiply of the same operands.  C's index arrays are
            # kept too, so that the passes see the same arrays every time.  (Round 3 cropped both operands and ran a full symbolic phase
            # per pass and multiply: 150 ms of config 5's 2018 ms, profiles/r04_config5_step_breakdown.txt.)
            # The views and the passes' engines depend on A and B only (with C's stamp in this key the usual in / out pattern --
            # the previous result passed back as Cm, fresh index tensors every call -- never repeated a key, and every multiply rebuilt all
            # views and made n new engines with work areas of tens of GB).  A new C pattern meets the old engines: their device-side plan
            # comparison rebuilds a pass's plan in place only when the arrays really differ.
            key = (A.index_stamp(), B.index_stamp(), n, bool(retain_sparsity))
            if getattr(self, "_kpass_key", None) != key:
                self._kpass = None   # (the old passes' engines and their work areas go first)
                self._kpass_key, self._kpass = key, self._kpass_views(A, B, n, total)
                self._kpass_cidx = None
            # What the passes accumulate into: ONE wrapper object over PRIVATE copies of C_out's index arrays, kept while C_in and the operands
            # are the same generation of the same arrays -- the passes then see the same arrays with the same stamp every time and their
            # trusted plans engage.  The copies are this cache's own: the matrix handed back to the caller keeps the index tensors init_c
            # made for it, so nothing a caller does to a result can reach a later multiply (and the library's rewrite of the cached col_i /
            # blk_p in every pass touches no earlier result).  The version counters guard against a torch-side edit all the same.
            ckey = (Cm.index_stamp(), key, int(out.col_i.numel()))
            c = getattr(self, "_kpass_cidx", None)
            if c is not None and c["key"] == ckey and all(t._version == v for t, v in zip(c["tensors"], c["versions"])):
                work = c["work"]
                work.data = out.data
            else:
                # (the copies are made on the stream init_c wrote the arrays on: on torch's current stream they could read col_i / blk_p of a
                # caller's side stream before init_c has filled them, and the trusted pass plans would keep the half-filled index)
                with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
                    work = DbcsrMatrix(out.row_blk_size, out.col_blk_size, out.row_p.clone(), out.col_i.clone(), out.blk_p.clone(), out.data, out.name)
                tensors = (work.row_p, work.col_i, work.blk_p)
                self._kpass_cidx = {"key": ckey, "work": work, "tensors": tensors, "versions": tuple(t._version for t in tensors)}
            flop = nprod = 0
            self.pass_launches = []   # [(kernel ms, flop)] per pass, filled when collect_kernel_times is set (a synchronisation per pass)
            for Ac, Bc, eng_c in self._kpass:
                Ac.data, Bc.data = A.data, B.data   # (the values may be new ones: same arrays or not, the views follow)
                cnt = eng_c.accumulate(alpha, Ac, Bc, work, stream=stream)
                if getattr(self, "collect_kernel_times", False):
                    self.pass_launches.append((float(MultiplyEngine._last_timing(eng_c)[1]), int(cnt.flop)))
                flop += cnt.flop
                nprod += cnt.nproducts
                self.last_launch_flop, self.last_kchunks = cnt.flop, n  # what last_timing() refers to
                self._last_pass_engine = eng_c
            total.flop, total.nproducts = flop, nprod
            return out, total
        st = StreamHandle(stream)
        dev = A.data.device
        a, b, cin = A.desc(), B.desc(), Cm.desc()
        row_p = torch.empty(Cm.nblkrows + 1, dtype=torch.int32, device=dev)
        counts = _lib.MmCounts()
        rc = self.L.dbcsr_amd_mm_symbolic_filtered(self.h, A.dtype_code, float(alpha), float(filter_eps or 0.0), C.byref(a), C.byref(b),
                                                   C.byref(cin), 1 if retain_sparsity else 0, row_p.data_ptr(), C.byref(counts), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_mm_symbolic failed (%d)" % rc)
        out = DbcsrMatrix(Cm.row_blk_size, Cm.col_blk_size, row_p, torch.empty(counts.c_nblks, dtype=torch.int32, device=dev),
                          torch.empty(counts.c_nblks, dtype=torch.int64, device=dev),
                          torch.empty(counts.c_nze, dtype=A.dtype, device=dev), Cm.name)
        cout = out.desc(out=True)
        if filter_eps and filter_eps > 0 and not retain_sparsity and os.environ.get("DBCSR_AMD_MM_EXPECT_FILTER", "1") != "0":
            # the product goes straight into the block filter below with the same eps (dbcsr_mm_multrec.F:373-383): blocks it will drop need not be written
            self.L.dbcsr_amd_mm_expect_filter(self.h, float(filter_eps))
        rc = self.L.dbcsr_amd_mm_numeric(self.h, A.dtype_code, float(alpha), C.byref(a), C.byref(b), float(beta), C.byref(cin),
                                         C.byref(cout), st.ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_mm_numeric failed (%d)" % rc)
        self.last_launch_flop, self.last_kchunks = counts.flop, 1
        if filter_eps and filter_eps > 0 and not retain_sparsity:  # dbcsr_mm_multrec.F:373-383
            out = self.filtered(out, filter_eps, stream=stream)
        return out, counts


_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = MultiplyEngine()
    return _default_engine


def dbcsr_multiply(transa, transb, alpha, matrix_a, matrix_b, beta, matrix_c, first_row=None, last_row=None, first_column=None,
                   last_column=None, first_k=None, last_k=None, retain_sparsity=False, filter_eps=None, flop=None, engine=None):
    """Reference signature (src/dbcsr_api.F:1411-1433).  ``matrix_c`` is updated
    in place (its index/data tensors are replaced); ``flop`` may be a one-element
    list that receives the flop count, as the reference's optional INTENT(OUT)."""
    for t in (transa, transb):
        if t not in ("N", "T", "C"):
            raise ValueError("dbcsr_multiply: invalid transpose flag %r" % (t,))
    if matrix_a.dtype != matrix_b.dtype or matrix_a.dtype != matrix_c.dtype:
        raise TypeError("dbcsr_multiply: data types of A, B and C differ")
    E = engine or default_engine()
    c_symm = getattr(matrix_c, "symmetry", "N")
    if c_symm != "N":
        # Product matrix with symmetry (src/mm/dbcsr_mm.F:711-719): its index goes into canonical (checkerboard) form, only the blocks
        # stored in that form are computed (dbcsr_mm_csr.F:280-292), the result goes back to the stored triangle (row <= column).
        if c_symm not in ("S", "A"):
            raise ValueError("unsupported matrix symmetry %r (real data: 'N', 'S', 'A')" % (c_symm,))
        if any(v is not None and v != 0 for v in (first_row, last_row, first_column, last_column, first_k, last_k)):
            raise NotImplementedError("dbcsr_multiply: limits with a symmetric product matrix (the reference's tests run full limits only)")
        canon = E.twin_moved(matrix_c, 1, c_symm)
        E.set_canonical_product(True)
        try:
            counts = dbcsr_multiply(transa, transb, alpha, matrix_a, matrix_b, beta, canon, retain_sparsity=retain_sparsity,
                                    filter_eps=filter_eps, flop=flop, engine=E)
        finally:
            E.set_canonical_product(False)
        up = E.twin_moved(canon, 2, c_symm)
        matrix_c.row_p, matrix_c.col_i, matrix_c.blk_p, matrix_c.data = up.row_p, up.col_i, up.blk_p, up.data
        return counts
    matrix_a, matrix_b = E.desymmetrized(matrix_a), E.desymmetrized(matrix_b)
    A = E.transposed(matrix_a) if transa != "N" else matrix_a
    B = E.transposed(matrix_b) if transb != "N" else matrix_b
    if A.nblkcols != B.nblkrows or A.nblkrows != matrix_c.nblkrows or B.nblkcols != matrix_c.nblkcols:
        raise ValueError("dbcsr_multiply: incompatible block dimensions")
    limits = (first_row, last_row, first_column, last_column, first_k, last_k)
    # Submatrix selection (src/mm/dbcsr_mm.F:625-692, 1-based inclusive full-matrix indices, None/0 = not given): invalid limits
    # are an error (DBCSR_ABORT in the reference), the default values are "optimised away" exactly as there.
    fr = lr = fc = lc = fk = lk = 0
    window_keeps = False
    if any(v is not None and v != 0 for v in limits):
        nr, nc, nk = int(A.row_blk_size.sum()), int(B.col_blk_size.sum()), int(A.col_blk_size.sum())
        fr, lr, fc, lc, fk, lk = [int(v or 0) for v in limits]
        if fr < 0 or fr > nr or lr < 0 or lr > nr or (lr and fr > lr):
            raise ValueError("dbcsr_multiply: invalid row limits")
        if fc < 0 or fc > nc or lc < 0 or lc > nc or (lc and fc > lc):
            raise ValueError("dbcsr_multiply: invalid column limits")
        if fk < 0 or fk > nk or lk < 0 or lk > nk or (lk and fk > lk):
            raise ValueError("dbcsr_multiply: invalid k limits")
        fr, lr = (0 if fr == 1 else fr), (0 if lr == nr else lr)
        fc, lc = (0 if fc == 1 else fc), (0 if lc == nc else lc)
        fk, lk = (0 if fk == 1 else fk), (0 if lk == nk else lk)
        window_keeps = (0 < lc < nc) or (0 < lr < nr)
    limited = any((fr, lr, fc, lc, fk, lk))
    # Product data is retained when retain_sparsity, beta != 0, or a row / column window ends inside C (dbcsr_mm.F:695-704);
    # otherwise the reference empties C before the multiplication (:865-870): old blocks vanish, their values are never read.
    keep_product_data = bool(retain_sparsity) or beta != 0.0 or window_keeps
    matrix_in, beta_eff = matrix_c, beta
    if not keep_product_data:
        matrix_in, beta_eff = E.empty_like(matrix_c), 1.0
    if limited:
        # the left matrix is cropped to (rows, k), the right one to (k, columns) (make_m2s, dbcsr_mm_cannon.F:194-214), beta acts
        # on the window of C only (dbcsr_scale with limits) and everything outside the window stays as it is
        nr, nc, nk = int(A.row_blk_size.sum()), int(B.col_blk_size.sum()), int(A.col_blk_size.sum())
        rb, cb, kb = ((fr or 1) - 1, (lr or nr) - 1), ((fc or 1) - 1, (lc or nc) - 1), ((fk or 1) - 1, (lk or nk) - 1)
        A, B = E.cropped(A, rb, kb), E.cropped(B, kb, cb)
        if keep_product_data and beta != 1.0:
            matrix_in = E.scaled_window(matrix_c, beta, rb, cb)
        beta_eff = 1.0
    out, counts = E.multiply_local(alpha, A, B, beta_eff, matrix_in, retain_sparsity=retain_sparsity, filter_eps=filter_eps or 0.0)
    matrix_c.row_p, matrix_c.col_i, matrix_c.blk_p, matrix_c.data = out.row_p, out.col_i, out.blk_p, out.data
    if flop is not None:
        flop[:] = [counts.flop]
    return counts
