"""Submatrix limits of dbcsr_multiply in the oracle, pinned on the reference's own unit-test specification:
the limit cases of tests/dbcsr_unittest1.F:95-240 checked the way tests/dbcsr_test_multiply.F:585-755 does it
(dense GEMM on the window, everything outside the window unchanged, residual / ((|A|+|B|+|C|) n eps) <= 10)."""
import numpy as np
import pytest

from oracle import oracle as orc

# (name, (M, N, K), sparsities (A, B, C), retain_sparsity, alpha, beta, bs_m, bs_n, bs_k, limits) -- real parts of the
# reference's complex alpha/beta (the real(8) instantiation of the test uses them, dbcsr_test_multiply.F:300-330)
CASES = [
    ("ALPHA", (20, 20, 20), (0.5, 0.5, 0.5), True, -3.0, 0.0, [1, 4], [1, 4], [1, 4], (2, 6, 3, 7, 6, 7)),
    ("BETA", (20, 20, 20), (0.5, 0.5, 0.5), True, 1.0, 3.0, [1, 4], [1, 4], [1, 4], (2, 6, 3, 7, 6, 7)),
    ("LIMITS_COL_1", (50, 50, 50), (0.0, 0.0, 0.0), False, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (1, 50, 1, 20, 1, 50)),
    ("LIMITS_COL_2", (50, 50, 50), (0.0, 0.0, 0.0), False, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (1, 50, 9, 18, 1, 50)),
    ("LIMITS_COL_3", (50, 50, 50), (0.5, 0.5, 0.5), True, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (1, 50, 9, 18, 1, 50)),
    ("LIMITS_COL_4", (25, 50, 75), (0.5, 0.5, 0.5), True, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (1, 25, 9, 18, 1, 75)),
    ("LIMITS_K_1", (50, 50, 50), (0.0, 0.0, 0.0), False, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (1, 50, 1, 50, 1, 20)),
    ("LIMITS_K_2", (50, 50, 50), (0.0, 0.0, 0.0), False, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (1, 50, 1, 50, 9, 18)),
    ("LIMITS_K_3", (50, 50, 50), (0.5, 0.5, 0.5), True, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (1, 50, 1, 50, 9, 18)),
    ("LIMITS_K_4", (25, 50, 75), (0.5, 0.5, 0.5), True, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (1, 25, 1, 50, 9, 18)),
    ("LIMITS_MIX_1", (50, 50, 50), (0.0, 0.0, 0.0), False, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (9, 18, 11, 20, 1, 50)),
    ("LIMITS_MIX_2", (50, 50, 50), (0.0, 0.0, 0.0), False, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (1, 50, 9, 10, 11, 20)),
    ("LIMITS_MIX_3", (50, 50, 50), (0.0, 0.0, 0.0), False, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (9, 20, 1, 50, 11, 18)),
    ("LIMITS_MIX_4", (50, 50, 50), (0.0, 0.0, 0.0), False, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (11, 20, 11, 20, 13, 18)),
    ("LIMITS_MIX_5", (50, 50, 50), (0.5, 0.5, 0.5), True, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (11, 20, 11, 20, 13, 18)),
    ("LIMITS_MIX_6", (25, 50, 75), (0.5, 0.5, 0.5), True, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (11, 20, 11, 20, 13, 18)),
    ("LIMITS_MIX_7", (25, 50, 75), (0.5, 0.5, 0.5), True, 1.0, 0.0, [1, 2], [1, 2, 1, 3], [1, 3, 1, 2, 1, 0], (11, 20, 11, 20, 6, 10)),
    ("LIMITS_ROW_1", (50, 50, 50), (0.0, 0.0, 0.0), False, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (1, 20, 1, 50, 1, 50)),
    ("LIMITS_ROW_2", (50, 50, 50), (0.0, 0.0, 0.0), False, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (9, 18, 1, 50, 1, 50)),
    ("LIMITS_ROW_3", (50, 50, 50), (0.5, 0.5, 0.5), True, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (9, 18, 1, 50, 1, 50)),
    ("LIMITS_ROW_4", (25, 50, 75), (0.5, 0.5, 0.5), True, 1.0, 0.0, [1, 2], [1, 2], [1, 2], (9, 18, 1, 50, 1, 75)),
    # beyond the reference list: limits that cut blocks, beta != 0, new blocks allowed
    ("CUT_NEW", (40, 36, 44), (0.6, 0.6, 0.7), False, 0.5, 2.0, [1, 5, 1, 3], [1, 4], [1, 7, 1, 2], (7, 29, 6, 31, 10, 33)),
]


def limit_case_matrices(case, transa="N", transb="N"):
    _, (M, N, K), sp, _, _, _, bs_m, bs_n, bs_k, _ = case
    return orc.perf_case(M, N, K, sp[0], sp[1], sp[2], bs_m, bs_n, bs_k, transa, transb)


def dense_expected(case, A, B, Cm, transa="N", transb="N"):
    _, _, _, retain, alpha, beta, _, _, _, lim = case
    Ad = A.to_dense() if transa == "N" else A.to_dense().T
    Bd = B.to_dense() if transb == "N" else B.to_dense().T
    Cd = Cm.to_dense()
    r, c, k = slice(lim[0] - 1, lim[1]), slice(lim[2] - 1, lim[3]), slice(lim[4] - 1, lim[5])
    E = Cd.copy()
    E[r, c] = beta * Cd[r, c] + alpha * Ad[r, k] @ Bd[k, c]
    if retain:  # dbcsr_impose_sparsity: only the blocks C already has
        mask = np.zeros_like(E, dtype=bool)
        ro = np.concatenate([[0], np.cumsum(Cm.row_sizes)])
        co = np.concatenate([[0], np.cumsum(Cm.col_sizes)])
        rows = Cm.rows()
        for b in range(Cm.nblks):
            mask[ro[rows[b]]:ro[rows[b] + 1], co[Cm.col_i[b]]:co[Cm.col_i[b] + 1]] = True
        E = np.where(mask, E, 0.0)
    return Ad, Bd, Cd, E


def reference_criterion(Ad, Bd, Cd, E, got, n):
    """tests/dbcsr_test_multiply.F:741-748 (infinity norms as dlange('I'))."""
    eps = np.finfo(np.float64).eps
    norm = lambda X: np.abs(X).sum(axis=1).max() if X.size else 0.0
    residual = norm(E - got)
    return residual / ((norm(Ad) + norm(Bd) + norm(Cd)) * n * eps)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("trans", ["NN", "TN", "NT"])
def test_oracle_limits_match_dense_window(case, trans):
    transa, transb = trans[0], trans[1]
    if trans != "NN" and case[0] not in ("BETA", "LIMITS_MIX_6", "LIMITS_MIX_7", "CUT_NEW"):
        pytest.skip("transposes on a subset")
    A, B, Cm = limit_case_matrices(case, transa, transb)
    _, _, _, retain, alpha, beta, _, _, _, lim = case
    out, info = orc.multiply_limits(transa, transb, alpha, A, B, beta, Cm, lim, retain_sparsity=retain)
    Ad, Bd, Cd, E = dense_expected(case, A, B, Cm, transa, transb)
    n = lim[3] - lim[2] + 1
    assert reference_criterion(Ad, Bd, Cd, E, out.to_dense(), n) <= 10.0
    if retain:
        assert np.array_equal(out.row_p, Cm.row_p) and np.array_equal(out.col_i, Cm.col_i)
    # columns ascending inside each row (dbcsr_finalize order)
    for r in range(out.nbr):
        cols = out.col_i[out.row_p[r]:out.row_p[r + 1]]
        assert np.all(np.diff(cols) > 0)


def test_crop_keeps_intersecting_blocks_only():
    A = orc.make_random_matrix(orc.make_block_sizes(30, [1, 4]), orc.make_block_sizes(26, [1, 3]), 0.3, 77)
    Cc = orc.crop(A, (5, 21), (4, 19))
    D = A.to_dense()
    E = np.zeros_like(D)
    E[5:22, 4:20] = D[5:22, 4:20]
    assert np.array_equal(Cc.to_dense(), E)
    ro = np.concatenate([[0], np.cumsum(A.row_sizes)])
    co = np.concatenate([[0], np.cumsum(A.col_sizes)])
    rows = Cc.rows()
    for b in range(Cc.nblks):
        r, c = rows[b], Cc.col_i[b]
        assert ro[r + 1] - 1 >= 5 and ro[r] <= 21 and co[c + 1] - 1 >= 4 and co[c] <= 19
