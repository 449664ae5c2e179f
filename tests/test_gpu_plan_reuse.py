"""Plan reuse of the multiply engine (include/dbcsr_amd_mm.h, dbcsr_amd_mm_plan_stats): a multiply whose operands have the index
arrays of the previous one skips its symbolic phase; anything else (another pattern, another block offset, another block size,
retain_sparsity, a filter, an intervening crop / filter / transpose on the same engine) must not.  Results are compared with the
CPU oracle every time, and the reused multiply must reproduce the first one bit for bit."""
import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, rel_err, to_dev

pytestmark = pytest.mark.gpu

H2O = (23 * 20 + 16, 23 * 18 + 16, 23 * 22 + 16, 0.6, 0.6, 0.7, [1, 23], [1, 23], [1, 23])
MIXED = (300, 280, 260, 0.5, 0.5, 0.6, [1, 13, 1, 23, 1, 32, 1, 7], [1, 23, 1, 5, 1, 32], [1, 13, 1, 32, 1, 9])
TINY = (240, 240, 240, 0.7, 0.7, 0.7, [1, 4], [1, 4, 1, 3], [1, 4, 1, 2])


def check(eng, A, B, Cm, alpha=0.7, beta=1.3, retain=False, filter_eps=None):
    ref, info = O.multiply("N", "N", alpha, A, B, beta, Cm, retain_sparsity=retain, filter_eps=filter_eps or 0.0)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", alpha, dA, dB, beta, dC, retain_sparsity=retain, filter_eps=filter_eps, engine=eng)
    torch.cuda.synchronize()
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert rel_err(out.data, ref.data) <= 1e-10
    return out


@pytest.mark.parametrize("case,env", [(H2O, {}), (MIXED, {}), (MIXED, {"DBCSR_AMD_MM_CLASSES": "2"}), (TINY, {}), (H2O, {"DBCSR_AMD_MM_TILE": "2"})],
                         ids=["h2o", "mixed", "mixed_classes", "tiny", "h2o_tile"])
def test_same_index_reuses_plan_and_reproduces(monkeypatch, case, env):
    for k in ("DBCSR_AMD_MM_PLAN", "DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_TILE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = MultiplyEngine(lab="DBCSR_AMD_MM_TILE" in env)   # (the tile dataflow lives in the lab build)
    A, B, Cm = O.perf_case(*case)
    first = check(eng, A, B, Cm)
    assert eng.plan_stats() == (0, 1)
    again = check(eng, A, B, Cm)
    assert eng.plan_stats() == (1, 1)
    assert np.array_equal(first.data, again.data), "the reused plan must give the same sums in the same order"
    # new values, same pattern: still the same plan
    A2 = O.Bcsr(A.row_sizes, A.col_sizes, A.row_p, A.col_i, A.blk_p, A.data * 1.5 - 0.25)
    check(eng, A2, B, Cm, alpha=-1.1, beta=0.4)
    assert eng.plan_stats() == (2, 1)


def test_changed_index_builds_a_new_plan(monkeypatch):
    monkeypatch.delenv("DBCSR_AMD_MM_PLAN", raising=False)
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*H2O)
    check(eng, A, B, Cm)
    # same number of blocks, one block of B moved to another column of its row
    col_i = B.col_i.copy()
    moved = False
    for r in range(len(B.row_sizes)):
        b0, b1 = B.row_p[r], B.row_p[r + 1]
        present = set(int(c) for c in col_i[b0:b1])
        free = [c for c in range(len(B.col_sizes) - 1) if c not in present and B.col_sizes[c] == 23]
        if b1 > b0 and free and B.col_sizes[col_i[b1 - 1]] == 23:
            cand = [c for c in free if c > (col_i[b1 - 2] if b1 - b0 > 1 else -1)]
            if cand:
                col_i[b1 - 1] = cand[-1] if cand[-1] > col_i[b1 - 1] else col_i[b1 - 1]
                moved = moved or col_i[b1 - 1] != B.col_i[b1 - 1]
                if moved:
                    break
    assert moved
    B2 = O.Bcsr(B.row_sizes, B.col_sizes, B.row_p, col_i, B.blk_p, B.data)
    check(eng, A, B2, Cm)
    assert eng.plan_stats() == (0, 2)
    # retain_sparsity is part of the plan
    check(eng, A, B2, Cm, alpha=1.0, beta=1.0, retain=True)
    assert eng.plan_stats() == (0, 3)
    check(eng, A, B2, Cm, alpha=1.0, beta=1.0, retain=True)
    assert eng.plan_stats() == (1, 3)
    # a filtered multiply never reuses, and what follows it builds its own plan
    check(eng, A, B2, Cm, filter_eps=1e-3)
    check(eng, A, B2, Cm)
    reused, built = eng.plan_stats()
    assert reused == 1 and built >= 5


def test_plan_switch_off(monkeypatch):
    monkeypatch.setenv("DBCSR_AMD_MM_PLAN", "0")
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*H2O)
    check(eng, A, B, Cm)
    check(eng, A, B, Cm)
    assert eng.plan_stats() == (0, 2)


def test_trusted_plan_reuses_by_address_and_compares_everything_else(monkeypatch):
    """dbcsr_amd_mm_trust_plan: the same device arrays -> plan reused without a comparison (new VALUES in the same buffers are new
    values); other arrays, even with the same content, are compared as always; a changed pattern at another address builds a new plan"""
    monkeypatch.delenv("DBCSR_AMD_MM_PLAN", raising=False)
    eng = MultiplyEngine()
    eng.trust_plan(True)
    A, B, Cm = O.perf_case(*H2O)
    ref, _ = O.multiply("N", "N", 0.7, A, B, 1.3, Cm)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)

    def run(a, b, c, alpha, beta, expect):
        out, counts = eng.multiply_local(alpha, a, b, beta, c)
        torch.cuda.synchronize()
        got = dev_to_bcsr(out)
        assert np.array_equal(got.col_i, expect.col_i) and np.array_equal(got.blk_p, expect.blk_p)
        assert rel_err(got.data, expect.data) <= 1e-10

    run(dA, dB, dC, 0.7, 1.3, ref)
    run(dA, dB, dC, 0.7, 1.3, ref)
    assert eng.plan_stats() == (1, 1)
    dA.data.mul_(2.0)   # new values in place: same plan, new product
    ref2, _ = O.multiply("N", "N", 0.7, O.Bcsr(A.row_sizes, A.col_sizes, A.row_p, A.col_i, A.blk_p, A.data * 2.0), B, 1.3, Cm)
    run(dA, dB, dC, 0.7, 1.3, ref2)
    assert eng.plan_stats() == (2, 1)
    # the same index at OTHER addresses: compared, found equal, reused
    dB2 = to_dev(B)
    run(dA, dB2, dC, 0.7, 1.3, ref2)
    assert eng.plan_stats() == (3, 1)
    # another pattern of C_in (at other addresses): a new plan
    A3, B3, C3 = O.perf_case(H2O[0], H2O[1], H2O[2], 0.6, 0.6, 0.5, [1, 23], [1, 23], [1, 23])
    ref3, _ = O.multiply("N", "N", 0.7, O.Bcsr(A.row_sizes, A.col_sizes, A.row_p, A.col_i, A.blk_p, A.data * 2.0), B, 1.3, C3)
    run(dA, dB2, to_dev(C3), 0.7, 1.3, ref3)
    assert eng.plan_stats() == (3, 2)


def test_trusted_plan_is_not_fooled_by_a_reallocation_at_the_same_address(monkeypatch):
    """An operand that is freed and allocated again AT THE SAME ADDRESSES with the same block counts and another pattern: trust by
    address alone would multiply it with the old plan.  The index stamp of the new matrix differs, so its index is
    compared on the device and a fresh plan is built; a matrix whose index the library has just rewritten in place gets a new stamp too."""
    monkeypatch.delenv("DBCSR_AMD_MM_PLAN", raising=False)
    eng = MultiplyEngine()
    eng.trust_plan(True)
    A, B, Cm = O.perf_case(*H2O)
    # A' = A with its block columns mirrored inside every block row: the same number of blocks per row (row_p identical), the same block sizes
    # (uniform columns except the tail, which stays where it is), another pattern
    nbc = len(A.col_sizes)
    perm = np.arange(nbc)
    perm[:nbc - 1] = perm[:nbc - 1][::-1]
    col2 = A.col_i.copy()
    for r in range(len(A.row_sizes)):
        seg = np.sort(perm[A.col_i[A.row_p[r]:A.row_p[r + 1]]])
        col2[A.row_p[r]:A.row_p[r + 1]] = seg
    blk2 = np.zeros_like(A.blk_p)
    off = 0
    for r in range(len(A.row_sizes)):
        for ib in range(A.row_p[r], A.row_p[r + 1]):
            blk2[ib] = off
            off += int(A.row_sizes[r]) * int(A.col_sizes[col2[ib]])
    assert not np.array_equal(col2, A.col_i)
    A2 = O.Bcsr(A.row_sizes, A.col_sizes, A.row_p, col2, blk2, np.resize(A.data, off) if off > A.data.size else A.data[:off].copy())
    ref1, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
    ref2, _ = O.multiply("N", "N", 1.0, A2, B, 1.0, Cm)
    dB, dC = to_dev(B), to_dev(Cm)
    dA = to_dev(A)
    ptrs = (dA.row_p.data_ptr(), dA.col_i.data_ptr(), dA.blk_p.data_ptr())
    stamp1 = dA.index_stamp()

    def run(a, expect):
        out, _ = eng.multiply_local(1.0, a, dB, 1.0, dC)
        torch.cuda.synchronize()
        got = dev_to_bcsr(out)
        assert np.array_equal(got.col_i, expect.col_i) and rel_err(got.data, expect.data) <= 1e-10

    run(dA, ref1)
    run(dA, ref1)
    assert eng.plan_stats() == (1, 1)
    del dA
    torch.cuda.synchronize()
    dA2 = to_dev(A2)   # torch's caching allocator hands the freed blocks out again
    if (dA2.row_p.data_ptr(), dA2.col_i.data_ptr(), dA2.blk_p.data_ptr()) != ptrs:
        pytest.skip("the allocator did not reuse the addresses (nothing to be fooled by)")
    assert dA2.index_stamp() != stamp1
    run(dA2, ref2)
    assert eng.plan_stats() == (1, 2)   # compared on the device, found different: a fresh plan
    run(dA2, ref2)
    assert eng.plan_stats() == (2, 2)   # and THAT plan is reused by address
    # torch writes an index tensor in place: the stamp moves with its version counter
    s = dA2.index_stamp()
    dA2.col_i.add_(0)
    assert dA2.index_stamp() != s
