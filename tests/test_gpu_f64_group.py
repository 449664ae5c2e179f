"""The fp64 group kernel (dbcsr_amd/csrc/mm_group64.hip: a wave owns R C blocks of one block column, walks the merged list of their
products and shares every B block among them) against the CPU oracle AND, bit for bit, against the one-wave-per-block kernel (the
summation order per C block is the same by construction): R = 2 ... 6 forced (DBCSR_AMD_MM_F64_GROUP), cubes of 23 and 16, dense and
sparse C (groups with empty slots, lists of very different lengths, C blocks without products), tail blocks in EVERY dimension -- the
inner tail is BASELINE config 2's own shape (32768 = 1424 x 23 + 16) --, row counts that are no multiple of R or of 8 R, several column
panels, retain_sparsity with in-place accumulation, plan reuse (tables and merged lists kept), and the cases where the kernel must stand
back: a B whose blocks do not lie in index order, a filtered multiply.  Values 1e-10 relative (the bar of north_star), index bit-exact.
The kernel lost to the one-wave-per-block kernel on MI355X (profiles/r06_f64_group_kernel.txt) and lives in the LAB build: every engine here loads it."""
import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, to_dev

pytestmark = pytest.mark.gpu
ENV = ("DBCSR_AMD_MM_KERNEL", "DBCSR_AMD_MM_HOT", "DBCSR_AMD_MM_F64_GROUP", "DBCSR_AMD_MM_WG_WAVES", "DBCSR_AMD_MM_KCHUNKS", "DBCSR_AMD_MM_PANEL_MB",
       "DBCSR_AMD_MM_GROUP_PANEL_MB", "DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_SYMBOLIC", "DBCSR_AMD_MM_PLAN")

# name: (M, N, K, sparsity A, B, C, block size)
CASES = {
    "dense23": (23 * 19, 23 * 13, 23 * 17, 0.3, 0.3, 0.2, 23),
    "fill10_23": (23 * 61, 23 * 47, 23 * 80, 0.9, 0.9, 0.9, 23),           # config 2's fill: most steps hold one product
    "sparse23": (23 * 37, 23 * 21, 23 * 30, 0.85, 0.85, 0.9, 23),          # most groups have empty slots, many C blocks have no product
    "edges23": (23 * 18 + 16, 23 * 12 + 7, 23 * 16 + 16, 0.5, 0.5, 0.5, 23),  # tail block row, column AND inner tail (config 2's shape)
    "inner_tail23": (23 * 9, 23 * 11, 23 * 14 + 5, 0.4, 0.4, 0.6, 23),
    "one_row23": (23, 23 * 9, 23 * 11, 0.2, 0.4, 0.3, 23),
    "cube16": (16 * 41, 16 * 22, 16 * 35 + 9, 0.6, 0.6, 0.6, 16),
}
RS = {23: [2, 3, 4, 5, 6], 16: [2, 3, 4]}


def build(case):
    M, N, K, sa, sb, sc, bs = case
    return O.perf_case(M, N, K, sa, sb, sc, [1, bs], [1, bs], [1, bs])


def check(out, ref, tol=1e-10):
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i)
    if ref.data.size:
        scale = max(float(np.max(np.abs(ref.data))), 1e-300)
        assert float(np.max(np.abs(out.data - ref.data))) <= tol * scale


def clean(monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)


def run(A, B, Cm, alpha=0.75, beta=1.25, **kw):
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", alpha, dA, dB, beta, dC, flop=flop, engine=eng, **kw)
    torch.cuda.synchronize()
    return dev_to_bcsr(dC), eng.last_kernel(), flop[0]


@pytest.mark.parametrize("name,R", [(n, R) for n in sorted(CASES) for R in RS[CASES[n][6]]])
def test_group_kernel_matches_oracle_and_the_block_kernel_bitwise(monkeypatch, name, R):
    clean(monkeypatch)
    A, B, Cm = build(CASES[name])
    bs = CASES[name][6]
    ref, info = O.multiply("N", "N", 0.75, A, B, 1.25, Cm)
    plain, kname, _ = run(A, B, Cm)
    assert kname == "mm_numeric_f64_hot<%d,%d,%d>" % (bs, bs, bs), kname
    monkeypatch.setenv("DBCSR_AMD_MM_F64_GROUP", str(R))
    out, kname, flop = run(A, B, Cm)
    assert kname == "mm_numeric_f64_group<%d,%d,%d;%d>" % (bs, bs, bs, R), kname
    assert flop == info["flop"]
    check(out, ref)
    assert np.array_equal(out.blk_p, plain.blk_p) and np.array_equal(out.data, plain.data), "not the bits of the one-wave-per-block kernel"


@pytest.mark.parametrize("R", [2, 4, 6])
def test_group_kernel_panels_retain_and_in_place(monkeypatch, R):
    """several column panels (a tiny panel size), retain_sparsity, then a second product accumulated into the result; one engine:
    the second multiply reuses the plan, the table and the merged lists"""
    clean(monkeypatch)
    monkeypatch.setenv("DBCSR_AMD_MM_F64_GROUP", str(R))
    monkeypatch.setenv("DBCSR_AMD_MM_GROUP_PANEL_MB", "1")
    A, B, Cm = build(CASES["fill10_23"])
    ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm, retain_sparsity=True)
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f64_group<23,23,23;%d>" % R, eng.last_kernel()
    check(dev_to_bcsr(dC), ref)
    ref2, _ = O.multiply("N", "N", -0.5, A, B, 1.0, ref, retain_sparsity=True)
    dbcsr_multiply("N", "N", -0.5, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f64_group<23,23,23;%d>" % R, eng.last_kernel()
    check(dev_to_bcsr(dC), ref2)


def test_group_kernel_plan_reuse_and_new_pattern(monkeypatch):
    """the same operands three times (the merged lists are built once), then operands with another pattern on the same engine"""
    clean(monkeypatch)
    monkeypatch.setenv("DBCSR_AMD_MM_F64_GROUP", "4")
    A, B, Cm = build(CASES["edges23"])
    ref, info = O.multiply("N", "N", 1.0, A, B, 0.0, Cm)
    eng = MultiplyEngine(lab=True)
    dA, dB = to_dev(A), to_dev(B)
    for _ in range(3):
        dC = to_dev(Cm)
        dbcsr_multiply("N", "N", 1.0, dA, dB, 0.0, dC, engine=eng)
        torch.cuda.synchronize()
        assert eng.last_kernel() == "mm_numeric_f64_group<23,23,23;4>", eng.last_kernel()
        check(dev_to_bcsr(dC), ref)
    A2, B2, C2 = build(CASES["sparse23"])
    ref2, _ = O.multiply("N", "N", 1.0, A2, B2, 1.0, C2)
    dA2, dB2, dC2 = to_dev(A2), to_dev(B2), to_dev(C2)
    dbcsr_multiply("N", "N", 1.0, dA2, dB2, 1.0, dC2, engine=eng)
    torch.cuda.synchronize()
    check(dev_to_bcsr(dC2), ref2)


def test_group_kernel_transposed_operands(monkeypatch):
    clean(monkeypatch)
    monkeypatch.setenv("DBCSR_AMD_MM_F64_GROUP", "4")
    M, N, K, sa, sb, sc, bs = CASES["dense23"]
    A, B, Cm = O.perf_case(M, N, K, sa, sb, sc, [1, bs], [1, bs], [1, bs], transa="T", transb="T")
    ref, _ = O.multiply("T", "T", 1.5, A, B, 0.5, Cm)
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("T", "T", 1.5, dA, dB, 0.5, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f64_group<23,23,23;4>", eng.last_kernel()
    check(dev_to_bcsr(dC), ref)


def test_group_kernel_stands_back(monkeypatch):
    clean(monkeypatch)
    monkeypatch.setenv("DBCSR_AMD_MM_F64_GROUP", "4")
    # (a) B's blocks placed in memory in REVERSE index order: within a column the offsets no longer ascend with k
    A, B, Cm = build(CASES["dense23"])
    ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
    nb = B.col_i.size
    sizes = (np.asarray(B.row_sizes)[np.repeat(np.arange(len(B.row_sizes)), np.diff(B.row_p))] * np.asarray(B.col_sizes)[B.col_i]).astype(np.int64)
    new_p = np.zeros(nb, np.int64)
    new_p[::-1] = np.concatenate([[0], np.cumsum(sizes[::-1])[:-1]])
    data = np.empty_like(B.data)
    for b in range(nb):
        data[new_p[b]:new_p[b] + sizes[b]] = B.data[B.blk_p[b]:B.blk_p[b] + sizes[b]]
    Brev = O.Bcsr(B.row_sizes, B.col_sizes, B.row_p, B.col_i, new_p, data)
    out, kname, _ = run(A, Brev, Cm, 1.0, 1.0)
    assert kname == "mm_numeric_f64_hot<23,23,23>", kname
    check(out, ref)
    # (b) a filtered multiply: the one-wave-per-block kernel leaves the block norms the final filter reads
    ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm, filter_eps=1e-3)
    out, kname, _ = run(A, B, Cm, 1.0, 1.0, filter_eps=1e-3)
    assert kname == "mm_numeric_f64_hot<23,23,23>", kname
    check(out, ref)
    # (c) a size without a group kernel
    A, B, Cm = O.perf_case(24 * 9, 24 * 8, 24 * 10, 0.5, 0.5, 0.5, [1, 24], [1, 24], [1, 24])
    ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
    out, kname, _ = run(A, B, Cm, 1.0, 1.0)
    assert kname == "mm_numeric_f64_hot<24,24,24>", kname
    check(out, ref)
    # (d) switched off
    monkeypatch.setenv("DBCSR_AMD_MM_F64_GROUP", "0")
    A, B, Cm = build(CASES["dense23"])
    ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
    out, kname, _ = run(A, B, Cm, 1.0, 1.0)
    assert kname == "mm_numeric_f64_hot<23,23,23>", kname
    check(out, ref)
