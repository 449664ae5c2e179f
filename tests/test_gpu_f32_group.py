"""The fp32 group kernel (dbcsr_amd/csrc/mm_group.hip: a wave owns R C blocks of one block column, walks the union of their product lists
in ascending k and shares every B block among them) against the CPU oracle: R = 2, 3, 4 forced (DBCSR_AMD_MM_F32_GROUP), cubes of 16, 24
and 32, dense and sparse C (groups with empty slots, lists of very different lengths, C blocks without products), block rows / columns
of another size at the edges (their C blocks go through the second launch), row counts that are no multiple of R or of 8 R (XCDs with
fewer groups), several column panels, retain_sparsity with in-place accumulation, k passes (each pass a launch of the group kernel with
its own table), and the cases where the kernel must stand back: an inner tail block, a B whose blocks do not lie in index order.
Round 6: the kernel lives in the LAB build (it does not win; dbcsr_amd/csrc/Makefile), every engine here loads it.
Values 2e-5 of the largest element (fp32 sums in another order than the oracle's), index bit-exact."""
import numpy as np
import pytest
import torch

from dbcsr_amd.matrix import DbcsrMatrix
from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, to_dev

pytestmark = pytest.mark.gpu
ENV = ("DBCSR_AMD_MM_KERNEL", "DBCSR_AMD_MM_HOT", "DBCSR_AMD_MM_F32_DIRECT", "DBCSR_AMD_MM_F32_GROUP", "DBCSR_AMD_MM_WG_WAVES", "DBCSR_AMD_MM_KCHUNKS",
       "DBCSR_AMD_MM_PANEL_MB", "DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_SYMBOLIC")

# name: (M, N, K, sparsity A, B, C, block size)
CASES = {
    "dense32": (32 * 19, 32 * 13, 32 * 17, 0.3, 0.3, 0.2, 32),
    "sparse32": (32 * 37, 32 * 21, 32 * 30, 0.85, 0.85, 0.9, 32),          # most groups have empty slots, many C blocks have no product
    "fill20_32": (32 * 40, 32 * 33, 32 * 48, 0.8, 0.8, 0.8, 32),           # config 5's fill
    "edges32": (32 * 18 + 20, 32 * 12 + 7, 32 * 16, 0.5, 0.5, 0.5, 32),    # tail block row and column (no inner tail)
    "one_row32": (32, 32 * 9, 32 * 11, 0.2, 0.4, 0.3, 32),
    "cube16": (16 * 41, 16 * 22, 16 * 35, 0.6, 0.6, 0.6, 16),
    "cube24": (24 * 23 + 5, 24 * 17, 24 * 19, 0.6, 0.6, 0.7, 24),
}


def cast32(Mx):
    return O.Bcsr(Mx.row_sizes, Mx.col_sizes, Mx.row_p, Mx.col_i, Mx.blk_p, Mx.data.astype(np.float32))


def wide(Mx):
    return O.Bcsr(Mx.row_sizes, Mx.col_sizes, Mx.row_p, Mx.col_i, Mx.blk_p, Mx.data.astype(np.float64))


def build(case):
    M, N, K, sa, sb, sc, bs = case
    A, B, Cm = O.perf_case(M, N, K, sa, sb, sc, [1, bs], [1, bs], [1, bs])
    return cast32(A), cast32(B), cast32(Cm)


def check(out, ref, tol=2e-5):
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i)
    if ref.data.size:
        scale = max(float(np.max(np.abs(ref.data))), 1e-300)
        assert float(np.max(np.abs(out.data.astype(np.float64) - ref.data))) <= tol * scale


@pytest.mark.parametrize("R", [2, 3, 4])
@pytest.mark.parametrize("name", sorted(CASES))
def test_group_kernel_matches_oracle(monkeypatch, name, R):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DBCSR_AMD_MM_F32_GROUP", str(R))
    A, B, Cm = build(CASES[name])
    bs = CASES[name][6]
    ref, info = O.multiply("N", "N", 0.75, wide(A), wide(B), 1.25, wide(Cm))
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", 0.75, dA, dB, 1.25, dC, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f32_group<%d,%d,%d;%d>" % (bs, bs, bs, R), eng.last_kernel()
    assert flop[0] == info["flop"]
    check(dev_to_bcsr(dC), ref)


@pytest.mark.parametrize("R", [2, 4])
def test_group_kernel_panels_retain_and_in_place(monkeypatch, R):
    """several column panels (a tiny panel size), retain_sparsity, then a second product accumulated into the result"""
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DBCSR_AMD_MM_F32_GROUP", str(R))
    monkeypatch.setenv("DBCSR_AMD_MM_PANEL_MB", "1")
    A, B, Cm = build(CASES["fill20_32"])
    ref, _ = O.multiply("N", "N", 1.0, wide(A), wide(B), 1.0, wide(Cm), retain_sparsity=True)
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("mm_numeric_f32_group<32,32,32;%d>" % R), eng.last_kernel()
    check(dev_to_bcsr(dC), ref)
    ref2, _ = O.multiply("N", "N", -0.5, wide(A), wide(B), 1.0, ref, retain_sparsity=True)
    dbcsr_multiply("N", "N", -0.5, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
    torch.cuda.synchronize()
    check(dev_to_bcsr(dC), ref2)


@pytest.mark.parametrize("npass", [2, 3])
def test_group_kernel_in_k_passes(monkeypatch, npass):
    """config 5's mode: passes over k ranges accumulating in place, every pass a launch of the group kernel with a plan of its own;
    repeated, so that the plans and the group tables are reused"""
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DBCSR_AMD_MM_F32_GROUP", "4")
    monkeypatch.setenv("DBCSR_AMD_MM_KCHUNKS", str(npass))
    A, B, Cm = build((32 * 21, 32 * 15, 32 * 96, 0.8, 0.8, 0.7, 32))
    ref, info = O.multiply("N", "N", 1.0, wide(A), wide(B), 1.0, wide(Cm))
    eng = MultiplyEngine(lab=True)
    dA, dB = to_dev(A), to_dev(B)
    for _ in range(3):
        dC = to_dev(Cm)
        flop = [0]
        dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, flop=flop, engine=eng)
        torch.cuda.synchronize()
        assert eng.last_kchunks == npass and eng.last_kernel() == "mm_numeric_f32_group<32,32,32;4>", (eng.last_kchunks, eng.last_kernel())
        assert flop[0] == info["flop"]
        check(dev_to_bcsr(dC), ref)


def test_group_kernel_stands_back(monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DBCSR_AMD_MM_F32_GROUP", "4")
    # (a) an inner tail block: products of another k extent would sit in the lists
    A, B, Cm = build((32 * 12, 32 * 11, 32 * 13 + 12, 0.5, 0.5, 0.5, 32))
    ref, _ = O.multiply("N", "N", 1.0, wide(A), wide(B), 1.0, wide(Cm))
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f32_direct<32,32,32>", eng.last_kernel()
    check(dev_to_bcsr(dC), ref)
    # (b) B's blocks placed in memory in REVERSE index order: within a column the offsets no longer ascend with k
    A, B, Cm = build(CASES["dense32"])
    ref, _ = O.multiply("N", "N", 1.0, wide(A), wide(B), 1.0, wide(Cm))
    nb = B.col_i.size
    sizes = (np.asarray(B.row_sizes)[np.repeat(np.arange(len(B.row_sizes)), np.diff(B.row_p))] * np.asarray(B.col_sizes)[B.col_i]).astype(np.int64)
    new_p = np.zeros(nb, np.int64)
    new_p[::-1] = np.concatenate([[0], np.cumsum(sizes[::-1])[:-1]])
    data = np.empty_like(B.data)
    for b in range(nb):
        data[new_p[b]:new_p[b] + sizes[b]] = B.data[B.blk_p[b]:B.blk_p[b] + sizes[b]]
    Brev = O.Bcsr(B.row_sizes, B.col_sizes, B.row_p, B.col_i, new_p, data)
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(Brev), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f32_direct<32,32,32>", eng.last_kernel()
    check(dev_to_bcsr(dC), ref)
    # (c) switched off
    monkeypatch.setenv("DBCSR_AMD_MM_F32_GROUP", "0")
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f32_direct<32,32,32>", eng.last_kernel()
    check(dev_to_bcsr(dC), ref)


def test_group_kernel_automatic_choice(monkeypatch):
    """DBCSR_AMD_MM_F32_GROUP=-1: long lists (at least 16 products per C block on average, 1024 blocks) take R = 4, short ones the
    one-wave-per-block kernel; unset: the one-wave-per-block kernel always (the group kernel does not win on MI355X, see mm_engine.hip)"""
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    A, B, Cm = build((32 * 36, 32 * 34, 32 * 120, 0.6, 0.6, 0.5, 32))   # 120 * 0.16 = 19 products per block
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f32_direct<32,32,32>", eng.last_kernel()
    monkeypatch.setenv("DBCSR_AMD_MM_F32_GROUP", "-1")
    ref, _ = O.multiply("N", "N", 1.0, wide(A), wide(B), 1.0, wide(Cm))
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f32_group<32,32,32;4>", eng.last_kernel()
    check(dev_to_bcsr(dC), ref)
    A, B, Cm = build(CASES["sparse32"])
    eng = MultiplyEngine(lab=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f32_direct<32,32,32>", eng.last_kernel()
