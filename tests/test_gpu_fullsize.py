"""Parity at BASELINE.json's FULL sizes through size-independent properties (the oracle finishes these sizes in minutes,
not seconds, so here the device result is checked against identities instead):
  * index structure bit-exact against an independent dense boolean product of the block patterns,
    columns ascending inside every row, blk_p = running sum of the block sizes;
  * (C0 + A B) x  ==  C0 x + A (B x) for a dense block vector x (three thin multiplies that run other kernels);
  * linearity of the position-dependent checksum: cs_pos(alpha A B + beta C0) = alpha cs_pos(A B) + beta cs_pos(C0);
  * transposition: checksum(B^T A^T) = checksum(A B), same number of blocks;
  * idempotence: alpha = 0, beta = 1, retain_sparsity leaves C bit-identical.
Tolerance 1e-10 relative (north star), structure exact."""
import pytest
import torch

from dbcsr_amd.matrix import DbcsrMatrix
from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from dbcsr_amd.randmat import make_random_matrix, perf_matrices

pytestmark = pytest.mark.gpu

CONFIGS = {
    "config1_4096_4x4_fill10": (4096, 0.10, [1, 4], torch.float64),
    "config2_32768_23x23_fill10": (32768, 0.10, [1, 23], torch.float64),
    "config3_32768_mixed_fill5": (32768, 0.05, [1, 13, 1, 23, 1, 32], torch.float64),
    "config4_131072_23x23_fill1": (131072, 0.01, [1, 23], torch.float64),
    # config 5's shape (32 x 32, 20 %, fp32) at a quarter of its edge: the full 131072^2 needs 143 GB and 2.7 s per multiply
    "config5_shape_32768_32x32_fill20_fp32": (32768, 0.20, [1, 32], torch.float32),
    # ... and at its full size on one GPU (180 TFLOP per multiply, 69 GB of C, 33 GB of product lists; four passes over k)
    "config5_131072_32x32_fill20_fp32": (131072, 0.20, [1, 32], torch.float32),
}


def pattern(M):
    nbr, nbc = M.nblkrows, M.nblkcols
    rows = torch.repeat_interleave(torch.arange(nbr, device=M.row_p.device), (M.row_p[1:] - M.row_p[:-1]).long())
    P = torch.zeros(nbr, nbc, dtype=torch.bool, device=M.row_p.device)
    P[rows, M.col_i.long()] = True
    return P, rows


def empty_like(rs, cs, dtype, dev):
    return DbcsrMatrix.empty_like_pattern(rs, cs, dtype, device=dev)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_full_size_properties(name):
    size, fill, mix, dtype = CONFIGS[name]
    tol = 1e-10 if dtype == torch.float64 else 2e-3  # fp32: sums of up to 32768 x 0.2 terms in two different orders
    E = MultiplyEngine()
    A, B, C0 = perf_matrices(size, size, size, (1.0 - fill,) * 3, mix, mix, mix, dtype=dtype, engine=E)
    dev = A.data.device
    out, counts = E.multiply_local(1.0, A, B, 1.0, C0)
    torch.cuda.synchronize()

    # ---- structure, bit-exact --------------------------------------------------------------------------------
    PA, _ = pattern(A)
    PB, _ = pattern(B)
    P0, _ = pattern(C0)
    expect = (PA.to(torch.float32) @ PB.to(torch.float32) > 0.5) | P0
    got, rows = pattern(out)
    assert torch.equal(got, expect)
    assert int(out.row_p[-1]) == out.col_i.numel() == int(expect.sum()) == counts.c_nblks
    same_row = rows[1:] == rows[:-1]
    assert bool(torch.all(out.col_i[1:][same_row] > out.col_i[:-1][same_row]))  # ascending inside a row, no duplicates
    nze = out.row_blk_size.long()[rows] * out.col_blk_size.long()[out.col_i.long()]
    assert torch.equal(out.blk_p, torch.cumsum(nze, 0) - nze)
    assert int(nze.sum()) == out.data.numel() == counts.c_nze
    # products and flops counted the way dbcsr_mm_csr.F:350 does
    nprod = (PA.to(torch.float64).sum(0) * PB.to(torch.float64).sum(1)).sum()
    assert int(nprod) == counts.nproducts
    ksz = A.col_blk_size.to(torch.float64)
    m_k = (PA.to(torch.float64) * A.row_blk_size.to(torch.float64)[:, None]).sum(0)   # sum of m over the A blocks of column k
    n_k = (PB.to(torch.float64) * B.col_blk_size.to(torch.float64)[None, :]).sum(1)   # sum of n over the B blocks of row k
    assert int((2.0 * m_k * ksz * n_k).sum()) == counts.flop
    del PA, PB, P0, expect, got

    # ---- (C0 + A B) x == C0 x + A (B x) ------------------------------------------------------------------------
    one = torch.ones(1, dtype=torch.int32)
    X = make_random_matrix(B.col_blk_size.cpu().numpy(), one.numpy(), 0.0, 999, dtype=dtype, engine=E)
    yrs = A.row_blk_size
    y1, _ = E.multiply_local(1.0, out, X, 0.0, empty_like(yrs, X.col_blk_size, A.dtype, dev))
    bx, _ = E.multiply_local(1.0, B, X, 0.0, empty_like(B.row_blk_size, X.col_blk_size, A.dtype, dev))
    y2, _ = E.multiply_local(1.0, C0, X, 0.0, empty_like(yrs, X.col_blk_size, A.dtype, dev))
    y2, _ = E.multiply_local(1.0, A, bx, 1.0, y2)
    torch.cuda.synchronize()
    assert y1.col_i.numel() == y2.col_i.numel() == A.nblkrows and y1.data.numel() == y2.data.numel()
    err = torch.max(torch.abs(y1.data - y2.data) / torch.clamp(torch.abs(y2.data), min=1e-300))
    assert float(err) <= tol
    del out, y1, y2, bx, X, rows, nze, same_row  # config 4's C is 60 GB: one product matrix alive at a time

    # ---- checksum identities -----------------------------------------------------------------------------------
    ab, _ = E.multiply_local(1.0, A, B, 0.0, empty_like(A.row_blk_size, B.col_blk_size, A.dtype, dev))
    cs_ab, ab_nblks = E.checksum(ab), ab.col_i.numel()
    del ab
    lin, _ = E.multiply_local(2.0, A, B, -3.0, C0)
    cs_c0, cs_lin = E.checksum(C0), E.checksum(lin)
    del lin
    assert rel(cs_lin[1], 2.0 * cs_ab[1] - 3.0 * cs_c0[1]) <= tol
    D = empty_like(B.col_blk_size, A.row_blk_size, A.dtype, dev)
    dbcsr_multiply("T", "T", 1.0, B, A, 0.0, D, engine=E)
    cs_d = E.checksum(D)
    assert D.col_i.numel() == ab_nblks
    assert rel(cs_d[0], cs_ab[0]) <= tol
    del D

    # ---- idempotence -------------------------------------------------------------------------------------------
    same, _ = E.multiply_local(0.0, A, B, 1.0, C0, retain_sparsity=True)
    assert torch.equal(same.row_p, C0.row_p) and torch.equal(same.col_i, C0.col_i) and torch.equal(same.data, C0.data)
