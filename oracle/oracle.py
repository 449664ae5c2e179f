"""ctypes front-end of oracle/liboracle.so (C restatement, see dbcsr_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dbcsr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def usable_cpus():
    """CPUs this process may run on at once: visible CPUs, affinity mask, cgroup quota (cpu.max) -- whichever is smallest"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        if "OMP_NUM_THREADS" not in os.environ:   # (an explicit setting stands)
            L.orc_set_threads.argtypes = [C.c_int]
            L.orc_set_threads(usable_cpus())
        L.orc_dlarnv1.argtypes = [i32p, C.c_int64, f64p]
        L.orc_slarnv1.argtypes = [i32p, C.c_int64, f32p]
        L.orc_set_larnv_seed.argtypes = [C.c_int] * 5 + [i32p]
        L.orc_make_block_sizes.argtypes = [C.c_int, i32p, C.c_int, i32p, C.c_int]
        L.orc_make_block_sizes.restype = C.c_int
        L.orc_random_pattern.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, i32p, i32p, C.c_int64]
        L.orc_random_pattern.restype = C.c_int64
        for nm, fp in (("orc_fill_blocks_d", f64p), ("orc_fill_blocks_s", f32p)):
            getattr(L, nm).argtypes = [C.c_int64, i32p, i32p, C.c_int, C.c_int, C.c_int, i32p, i32p, i64p, fp]
        for nm, fp in (("orc_checksum_d", f64p), ("orc_checksum_s", f32p)):
            f = getattr(L, nm)
            f.argtypes = [C.c_int, C.c_int, i32p, i32p, i32p, i32p, i64p, fp, C.c_int]
            f.restype = C.c_double
        L.orc_stack_calc_d.argtypes = [i32p, C.c_int, f64p, f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_stack_calc_s.argtypes = [i32p, C.c_int, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_stack7_calc_d.argtypes = [i32p, C.c_int, f64p, f64p, f64p]
        L.orc_transpose_d.argtypes = [i32p, C.c_int, f64p, C.c_int, C.c_int]
        L.orc_norms_d.argtypes = [f64p, C.c_int, i32p, i32p, f32p]
        L.orc_mat_init.argtypes = [f64p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_stack_init.argtypes = [i32p] + [C.c_int] * 7 + [C.c_uint, C.c_int]
        L.orc_check_sum.argtypes = [f64p, C.c_int64]
        L.orc_check_sum.restype = C.c_double
        mat = [C.c_int, C.c_int, i32p, i32p, i32p, i32p, i64p, f64p]
        L.orc_multiply_d.argtypes = ([C.c_char, C.c_char, C.c_double] + mat + mat + [C.c_double] + mat +
                                     [C.c_int, C.c_double, C.c_void_p, C.c_int])
        L.orc_multiply_d.restype = C.c_void_p
        for nm in ("orc_result_nblks", "orc_result_nze", "orc_result_flop", "orc_result_nproducts"):
            getattr(L, nm).argtypes = [C.c_void_p]
            getattr(L, nm).restype = C.c_int64
        L.orc_result_copy.argtypes = [C.c_void_p, i32p, i32p, i64p, f64p, i32p, i32p]
        L.orc_result_free.argtypes = [C.c_void_p]
        L.orc_multiply_rows_d.argtypes = [C.c_int, C.c_int, i32p, i32p, i32p, i32p, i64p, f64p, i32p, i32p, i32p, i64p,
                                          f64p, C.c_int, i32p, i32p, i64p, f64p, C.c_int]
        L.orc_multiply_rows_d.restype = C.c_int64
        L.orc_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


RANDMAT_SEED_INIT = 12341313  # src/ops/dbcsr_test_methods.F:75


class Bcsr:
    """Host BCSR matrix (0-based, 64-bit block offsets, column-major blocks)."""

    def __init__(self, row_sizes, col_sizes, row_p, col_i, blk_p, data):
        self.row_sizes = np.ascontiguousarray(row_sizes, np.int32)
        self.col_sizes = np.ascontiguousarray(col_sizes, np.int32)
        self.row_p = np.ascontiguousarray(row_p, np.int32)
        self.col_i = np.ascontiguousarray(col_i, np.int32)
        self.blk_p = np.ascontiguousarray(blk_p, np.int64)
        self.data = np.ascontiguousarray(data)

    @property
    def nbr(self):
        return len(self.row_sizes)

    @property
    def nbc(self):
        return len(self.col_sizes)

    @property
    def nblks(self):
        return len(self.col_i)

    def rows(self):
        return np.repeat(np.arange(self.nbr, dtype=np.int32), np.diff(self.row_p))

    def to_dense(self):
        ro = np.concatenate([[0], np.cumsum(self.row_sizes)])
        co = np.concatenate([[0], np.cumsum(self.col_sizes)])
        D = np.zeros((ro[-1], co[-1]), self.data.dtype)
        rows = self.rows()
        for b in range(self.nblks):
            r, c = rows[b], self.col_i[b]
            m, n = self.row_sizes[r], self.col_sizes[c]
            D[ro[r]:ro[r] + m, co[c]:co[c] + n] = self.data[self.blk_p[b]:self.blk_p[b] + m * n].reshape(n, m).T
        return D

    def _args(self):
        return [self.nbr, self.nbc, self.row_sizes, self.col_sizes, self.row_p, self.col_i, self.blk_p, self.data]


def dlarnv1(iseed, n):
    s = np.ascontiguousarray(iseed, np.int32).copy()
    x = np.empty(n, np.float64)
    lib().orc_dlarnv1(s, n, x)
    return x, s


def slarnv1(iseed, n):
    s = np.ascontiguousarray(iseed, np.int32).copy()
    x = np.empty(n, np.float32)
    lib().orc_slarnv1(s, n, x)
    return x, s


def larnv_seed(irow, nrow, icol, ncol, ival):
    s = np.zeros(4, np.int32)
    lib().orc_set_larnv_seed(irow, nrow, icol, ncol, ival, s)
    return s


def make_block_sizes(size_sum, mix):
    mix = np.ascontiguousarray(mix, np.int32)
    cap = size_sum + 1
    out = np.empty(cap, np.int32)
    n = lib().orc_make_block_sizes(size_sum, mix, len(mix) // 2, out, cap)
    return out[:n].copy()


def random_pattern(nrow, ncol, sparsity, counter):
    sp = sparsity / 100.0 if sparsity > 1 else sparsity
    cap = int(nrow * ncol * (1.0 - sp) * 1.2) + 1024
    while True:
        rows = np.empty(cap, np.int32)
        cols = np.empty(cap, np.int32)
        n = lib().orc_random_pattern(nrow, ncol, sparsity, counter, rows, cols, cap)
        if n <= cap:
            return rows[:n].copy(), cols[:n].copy()
        cap = n


def make_random_matrix(row_sizes, col_sizes, sparsity, counter, dtype=np.float64):
    """dbcsr_make_random_matrix (src/ops/dbcsr_test_methods.F:318-465), non-symmetric."""
    row_sizes = np.ascontiguousarray(row_sizes, np.int32)
    col_sizes = np.ascontiguousarray(col_sizes, np.int32)
    nrow, ncol = len(row_sizes), len(col_sizes)
    rows, cols = random_pattern(nrow, ncol, sparsity, counter)
    nze = row_sizes[rows].astype(np.int64) * col_sizes[cols].astype(np.int64)
    blk_p = np.concatenate([[0], np.cumsum(nze)[:-1]]).astype(np.int64) if len(nze) else np.zeros(0, np.int64)
    data = np.empty(int(nze.sum()), dtype)
    fill = lib().orc_fill_blocks_d if dtype == np.float64 else lib().orc_fill_blocks_s
    fill(len(rows), rows, cols, nrow, ncol, counter, row_sizes, col_sizes, blk_p, data)
    row_p = np.zeros(nrow + 1, np.int64)
    np.add.at(row_p, rows.astype(np.int64) + 1, 1)
    row_p = np.cumsum(row_p).astype(np.int32)
    return Bcsr(row_sizes, col_sizes, row_p, cols, blk_p, data)


def make_random_matrix_symmetric(sizes, sparsity, counter, symmetry):
    """dbcsr_make_random_matrix with symmetry = 'S' / 'A' (src/ops/dbcsr_test_methods.F:394-458): the candidates (row, col)
    of the geometric sequence left of the diagonal are dropped (dbcsr_get_stored_coordinates only names the owner, it does not
    move the block), the others are filled from their own seed; a diagonal block gets its (negative) transpose added.
    Returns the STORED (upper-triangle) matrix."""
    sizes = np.ascontiguousarray(sizes, np.int32)
    n = len(sizes)
    rows, cols = random_pattern(n, n, sparsity, counter)
    blocks = {}
    for r, c in zip(rows.tolist(), cols.tolist()):
        sr, sc = r + 1, c + 1
        if sc < sr:
            continue
        m, k = int(sizes[sr - 1]), int(sizes[sc - 1])
        v = np.empty(m * k, np.float64)
        lib().orc_fill_blocks_d(1, np.asarray([r], np.int32), np.asarray([c], np.int32), n, n, int(counter),
                                np.ascontiguousarray(np.where(np.arange(n) == r, m, sizes), np.int32),
                                np.ascontiguousarray(np.where(np.arange(n) == c, k, sizes), np.int32), np.zeros(1, np.int64), v)
        if sr == sc:
            blk = v.reshape(k, m).T  # [i][j] of the m x m block
            v = (blk + blk.T if symmetry == "S" else blk - blk.T).T.reshape(-1)
        blocks[(sr - 1, sc - 1)] = v
    keys = sorted(blocks)
    rr = np.asarray([kk[0] for kk in keys], np.int32)
    cc = np.asarray([kk[1] for kk in keys], np.int32)
    nze = sizes[rr].astype(np.int64) * sizes[cc].astype(np.int64) if len(keys) else np.zeros(0, np.int64)
    blk_p = np.concatenate([[0], np.cumsum(nze)[:-1]]).astype(np.int64) if len(keys) else np.zeros(0, np.int64)
    data = np.concatenate([blocks[kk] for kk in keys]) if keys else np.zeros(0)
    row_p = np.zeros(n + 1, np.int64)
    np.add.at(row_p, rr.astype(np.int64) + 1, 1)
    return Bcsr(sizes, sizes, np.cumsum(row_p).astype(np.int32), cc, blk_p, data)


def desymmetrize(M, symmetry):
    """dbcsr_desymmetrize_deep: the full matrix of a stored triangle (block (c, r) = +-block(r, c)^T)."""
    sign = 1.0 if symmetry == "S" else -1.0
    rows = M.rows()
    blocks = {}
    for b in range(M.nblks):
        r, c = int(rows[b]), int(M.col_i[b])
        m, n = int(M.row_sizes[r]), int(M.col_sizes[c])
        v = M.data[M.blk_p[b]:M.blk_p[b] + m * n]
        blocks[(r, c)] = v
        if r != c:
            blocks[(c, r)] = sign * v.reshape(n, m).T.reshape(-1)  # column-major n x m block of the transpose
    keys = sorted(blocks)
    rr = np.asarray([k[0] for k in keys], np.int32)
    cc = np.asarray([k[1] for k in keys], np.int32)
    nze = M.row_sizes[rr].astype(np.int64) * M.col_sizes[cc].astype(np.int64) if keys else np.zeros(0, np.int64)
    blk_p = np.concatenate([[0], np.cumsum(nze)[:-1]]).astype(np.int64) if keys else np.zeros(0, np.int64)
    data = np.concatenate([blocks[k] for k in keys]) if keys else np.zeros(0)
    row_p = np.zeros(M.nbr + 1, np.int64)
    np.add.at(row_p, rr.astype(np.int64) + 1, 1)
    return Bcsr(M.row_sizes, M.col_sizes, np.cumsum(row_p).astype(np.int32), cc, blk_p, data)


def checksum(M, pos=False):
    f = lib().orc_checksum_d if M.data.dtype == np.float64 else lib().orc_checksum_s
    return f(M.nbr, M.nbc, M.row_sizes, M.col_sizes, M.row_p, M.col_i, M.blk_p, M.data, 1 if pos else 0)


def checker_tr(row, col):
    """dbcsr_dist_operations.F:65-75 (1-based logical block coordinates): is the stored twin of block (row, col) the block (col, row)?"""
    return (((row + col) & 1) == 1) == (col >= row)


def move_to_twin(M, move, symmetry):
    """blocks (r, c) with move(r, c) become blocks (c, r) = +-block^T (0-based r, c); the others stay (square blocking)."""
    sign = 1.0 if symmetry == "S" else -1.0
    rows = M.rows()
    blocks = {}
    for b in range(M.nblks):
        r, c = int(rows[b]), int(M.col_i[b])
        m, n = int(M.row_sizes[r]), int(M.col_sizes[c])
        v = M.data[M.blk_p[b]:M.blk_p[b] + m * n]
        if move(r, c):
            blocks[(c, r)] = sign * v.reshape(n, m).T.reshape(-1)
        else:
            blocks[(r, c)] = v
    keys = sorted(blocks)
    rr = np.asarray([k[0] for k in keys], np.int32)
    cc = np.asarray([k[1] for k in keys], np.int32)
    nze = M.row_sizes[rr].astype(np.int64) * M.col_sizes[cc].astype(np.int64) if keys else np.zeros(0, np.int64)
    blk_p = np.concatenate([[0], np.cumsum(nze)[:-1]]).astype(np.int64) if keys else np.zeros(0, np.int64)
    data = np.concatenate([blocks[k] for k in keys]) if keys else np.zeros(0, M.data.dtype)
    row_p = np.zeros(M.nbr + 1, np.int64)
    np.add.at(row_p, rr.astype(np.int64) + 1, 1)
    return Bcsr(M.row_sizes, M.col_sizes, np.cumsum(row_p).astype(np.int32), cc, blk_p, data)


def multiply(transa, transb, alpha, A, B, beta, Cm, retain_sparsity=False, filter_eps=0.0, limits=None, c_symmetry=None):
    """C <- beta*C + alpha*op(A)*op(B); returns (C_out Bcsr, info dict).
    c_symmetry "S" / "A": Cm holds the stored triangle (row <= col) of a symmetric / antisymmetric product matrix.  The reference
    puts its index into canonical (checkerboard) form first (dbcsr_mm.F:711-719, dbcsr_make_index_canonical), computes only the
    blocks that are stored in that form (dbcsr_mm_csr.F:280-292) and goes back to the triangle at the end."""
    if c_symmetry in ("S", "A"):
        assert limits is None and np.array_equal(Cm.row_sizes, Cm.col_sizes)
        canon = move_to_twin(Cm, lambda r, c: r != c and checker_tr(r + 1, c + 1), c_symmetry)
        out, info = multiply(transa, transb, alpha, A, B, beta, canon, retain_sparsity, filter_eps, None, c_symmetry="canonical")
        return move_to_twin(out, lambda r, c: r > c, c_symmetry), info
    L = lib()
    lim = None
    if limits is not None:
        lim_arr = np.ascontiguousarray(limits, np.int32)
        lim = lim_arr.ctypes.data
    h = L.orc_multiply_d(transa.encode(), transb.encode(), float(alpha), *A._args(), *B._args(), float(beta),
                         *Cm._args(), 1 if retain_sparsity else 0, float(filter_eps), lim, 1 if c_symmetry == "canonical" else 0)
    if not h:
        raise ValueError("orc_multiply_d failed")
    try:
        nblks, nze = L.orc_result_nblks(h), L.orc_result_nze(h)
        row_p = np.empty(Cm.nbr + 1, np.int32)
        col_i = np.empty(nblks, np.int32)
        blk_p = np.empty(nblks, np.int64)
        data = np.empty(nze, np.float64)
        dr = np.empty(nblks, np.int32)
        dc = np.empty(nblks, np.int32)
        L.orc_result_copy(h, row_p, col_i, blk_p, data, dr, dc)
        info = dict(flop=L.orc_result_flop(h), nproducts=L.orc_result_nproducts(h), disc_row=dr, disc_col=dc)
    finally:
        L.orc_result_free(h)
    return Bcsr(Cm.row_sizes, Cm.col_sizes, row_p, col_i, blk_p, data), info


def stack_calc(stack, c, a, b, m, n, k, b_transposed):
    stack = np.ascontiguousarray(stack, np.int32)
    f = lib().orc_stack_calc_d if c.dtype == np.float64 else lib().orc_stack_calc_s
    f(stack, len(stack) // 3, c, a, b, m, n, k, 1 if b_transposed else 0)


def stack7_calc(stack7, c, a, b):
    stack7 = np.ascontiguousarray(stack7, np.int32)
    lib().orc_stack7_calc_d(stack7, len(stack7) // 7, c, a, b)


def transpose(trs_stack, data, m, n):
    trs_stack = np.ascontiguousarray(trs_stack, np.int32)
    lib().orc_transpose_d(trs_stack, len(trs_stack), data, m, n)


def norms(mat, offsets, nelems):
    offsets = np.ascontiguousarray(offsets, np.int32)
    nelems = np.ascontiguousarray(nelems, np.int32)
    out = np.empty(len(offsets), np.float32)
    lib().orc_norms_d(mat, len(offsets), offsets, nelems, out)
    return out


def mat_init(mat_n, x, y, seed):
    out = np.empty(mat_n * x * y, np.float64)
    lib().orc_mat_init(out, mat_n, x, y, seed)
    return out


def stack_init(nstack, nc, na, nb, m, n, k, rseed=1):
    # the reference's INIT_STACK (acc_bench.h:48-79) advances by nstack / nc + (rand() mod ... - ...) entries per C block: with fewer
    # than two entries per C block on average the step can stay zero for ever -- the restatement loops exactly as the original would
    if nstack < 2 * nc:
        raise ValueError("stack_init: needs nstack >= 2 * nc (the reference's generator does not terminate below that)")
    out = np.empty(3 * nstack, np.int32)
    lib().orc_stack_init(out, nstack, nc, na, nb, m, n, k, rseed, 1)
    return out


def multiply_rows(row_begin, row_end, A, B, Cm, nthreads=0):
    """Timed CPU baseline leg: rows [row_begin,row_end) of C += A*B into Cm.data."""
    return lib().orc_multiply_rows_d(row_begin, row_end, A.row_sizes, A.col_sizes, A.row_p, A.col_i, A.blk_p, A.data,
                                     B.col_sizes, B.row_p, B.col_i, B.blk_p, B.data, Cm.nbc, Cm.row_p, Cm.col_i,
                                     Cm.blk_p, Cm.data, nthreads)


def max_threads():
    return lib().orc_max_threads()


def perf_case(M, N, K, sp_a, sp_b, sp_c, bs_m, bs_n, bs_k, transa="N", transb="N", dtype=np.float64):
    """Inputs of the reference perf driver (tests/dbcsr_performance_multiply.F:323-413):
    C, then A, then B are created, the matrix counter being bumped before each."""
    sm, sn, sk = make_block_sizes(M, bs_m), make_block_sizes(N, bs_n), make_block_sizes(K, bs_k)
    c0 = RANDMAT_SEED_INIT
    Cm = make_random_matrix(sm, sn, sp_c, c0 + 1, dtype)
    A = make_random_matrix(sk, sm, sp_a, c0 + 2, dtype) if transa != "N" else make_random_matrix(sm, sk, sp_a, c0 + 2, dtype)
    B = make_random_matrix(sn, sk, sp_b, c0 + 3, dtype) if transb != "N" else make_random_matrix(sk, sn, sp_b, c0 + 3, dtype)
    return A, B, Cm


# ---- submatrix limits (dbcsr_multiply's first_row ... last_k) --------------------------------------------------
# Restated for small cases in numpy: dbcsr_crop_matrix (src/ops/dbcsr_operations.F:1652-1833: the blocks that
# intersect the bounds are copied, the parts of the boundary blocks outside the bounds are cleared), dbcsr_scale
# with limits (src/mm/dbcsr_mm.F:706-709) and the order of operations of dbcsr_multiply_generic
# (src/mm/dbcsr_mm.F:631-709; make_m2s crops the left matrix to (rows, k) and the right one to (k, columns),
# src/mm/dbcsr_mm_cannon.F:194-214).  Pinned by tests/test_oracle_limits.py against the dense check of the
# reference's own unit test (tests/dbcsr_test_multiply.F:585-755) on the limit cases of tests/dbcsr_unittest1.F.

def transposed(M):
    """dbcsr_new_transposed: block (r, c) of M becomes block (c, r), transposed."""
    rows = M.rows()
    order = np.lexsort((rows, M.col_i))  # sort by (new row = old column, new column = old row)
    nze = (M.row_sizes[rows[order]].astype(np.int64) * M.col_sizes[M.col_i[order]]) if M.nblks else np.zeros(0, np.int64)
    blk_p = np.concatenate([[0], np.cumsum(nze)[:-1]]).astype(np.int64) if M.nblks else np.zeros(0, np.int64)
    data = np.empty(int(nze.sum()), M.data.dtype)
    for t, b in enumerate(order):
        m, n = M.row_sizes[rows[b]], M.col_sizes[M.col_i[b]]
        blk = M.data[M.blk_p[b]:M.blk_p[b] + m * n].reshape(n, m)  # [col][row] of the m x n block
        data[blk_p[t]:blk_p[t] + m * n] = blk.T.reshape(-1)        # n x m block, column-major
    row_p = np.zeros(M.nbc + 1, np.int64)
    np.add.at(row_p, M.col_i.astype(np.int64) + 1, 1)
    return Bcsr(M.col_sizes, M.row_sizes, np.cumsum(row_p).astype(np.int32), rows[order], blk_p, data)


def _bounds(b, n):
    return (0, n - 1) if b is None else (int(b[0]), int(b[1]))


def crop(M, row_bounds=None, col_bounds=None):
    """Bounds are 0-based inclusive ELEMENT indices (None = everything)."""
    ro = np.concatenate([[0], np.cumsum(M.row_sizes)]).astype(np.int64)
    co = np.concatenate([[0], np.cumsum(M.col_sizes)]).astype(np.int64)
    r0, r1 = _bounds(row_bounds, ro[-1])
    c0, c1 = _bounds(col_bounds, co[-1])
    rows = M.rows()
    keep, chunks, off = [], [], 0
    for b in range(M.nblks):
        r, c = rows[b], M.col_i[b]
        m, n = int(M.row_sizes[r]), int(M.col_sizes[c])
        if ro[r] + m - 1 < r0 or ro[r] > r1 or co[c] + n - 1 < c0 or co[c] > c1:
            continue
        blk = M.data[M.blk_p[b]:M.blk_p[b] + m * n].reshape(n, m).copy()  # [col][row]
        gi = ro[r] + np.arange(m)
        gj = co[c] + np.arange(n)
        blk[:, (gi < r0) | (gi > r1)] = 0
        blk[(gj < c0) | (gj > c1), :] = 0
        keep.append(b)
        chunks.append(blk.reshape(-1))
    keep = np.asarray(keep, np.int64)
    row_p = np.zeros(M.nbr + 1, np.int64)
    if len(keep):
        np.add.at(row_p, rows[keep].astype(np.int64) + 1, 1)
    nze = np.asarray([len(x) for x in chunks], np.int64)
    blk_p = np.concatenate([[0], np.cumsum(nze)[:-1]]).astype(np.int64) if len(keep) else np.zeros(0, np.int64)
    data = np.concatenate(chunks) if chunks else np.zeros(0, M.data.dtype)
    return Bcsr(M.row_sizes, M.col_sizes, np.cumsum(row_p).astype(np.int32), M.col_i[keep] if len(keep) else np.zeros(0, np.int32),
                blk_p, data.astype(M.data.dtype))


def scale_window(M, beta, row_bounds=None, col_bounds=None):
    """dbcsr_scale(matrix, beta, limits): only the elements inside the bounds are scaled.  Returns a copy."""
    ro = np.concatenate([[0], np.cumsum(M.row_sizes)]).astype(np.int64)
    co = np.concatenate([[0], np.cumsum(M.col_sizes)]).astype(np.int64)
    r0, r1 = _bounds(row_bounds, ro[-1])
    c0, c1 = _bounds(col_bounds, co[-1])
    out = Bcsr(M.row_sizes, M.col_sizes, M.row_p, M.col_i, M.blk_p, M.data.copy())
    rows = M.rows()
    for b in range(M.nblks):
        r, c = rows[b], M.col_i[b]
        m, n = int(M.row_sizes[r]), int(M.col_sizes[c])
        blk = out.data[M.blk_p[b]:M.blk_p[b] + m * n].reshape(n, m)
        gi = ro[r] + np.arange(m)
        gj = co[c] + np.arange(n)
        mask = np.outer((gj >= c0) & (gj <= c1), (gi >= r0) & (gi <= r1))
        blk[mask] *= beta
    return out


def multiply_limits(transa, transb, alpha, A, B, beta, Cm, limits, retain_sparsity=False, filter_eps=0.0):
    """dbcsr_multiply with the reference's limits convention: (first_row, last_row, first_column, last_column,
    first_k, last_k), 1-based inclusive full-matrix indices, 0 = not given."""
    left = transposed(A) if transa.upper() != "N" else A
    right = transposed(B) if transb.upper() != "N" else B
    nr, nc, nk = int(left.row_sizes.sum()), int(right.col_sizes.sum()), int(left.col_sizes.sum())
    fr, lr, fc, lc, fk, lk = [int(x) for x in limits]
    # "optimise the default values away" and keep_product_data exactly as src/mm/dbcsr_mm.F:669-704
    fr, lr = (0 if fr == 1 else fr), (0 if lr == nr else lr)
    fc, lc = (0 if fc == 1 else fc), (0 if lc == nc else lc)
    fk, lk = (0 if fk == 1 else fk), (0 if lk == nk else lk)
    keep = bool(retain_sparsity) or beta != 0.0 or (0 < lc < nc) or (0 < lr < nr)
    rb = ((fr or 1) - 1, (lr or nr) - 1)
    cb = ((fc or 1) - 1, (lc or nc) - 1)
    kb = ((fk or 1) - 1, (lk or nk) - 1)
    if not keep:  # the product matrix is emptied before the multiplication (dbcsr_mm.F:865-870)
        Cs = Bcsr(Cm.row_sizes, Cm.col_sizes, np.zeros(Cm.nbr + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int64),
                  np.zeros(0, Cm.data.dtype))
    else:
        Cs = scale_window(Cm, beta, rb, cb) if beta != 1.0 else Cm
    return multiply("N", "N", alpha, crop(left, rb, kb), crop(right, kb, cb), 1.0, Cs, retain_sparsity=retain_sparsity,
                    filter_eps=filter_eps)
