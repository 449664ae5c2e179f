"""Multiplies whose block dimensions are all at most 8 (the range of libsmm_acc's tiny dataflow: src/acc/libsmm_acc/kernels/smm_acc_dnt_tiny.h;
5 x 5 x 5 ... 8 x 8 x 8 are tuned triplets of its parameter files): one wave per C block, ONE 8 x 8 tile, several products in flight
(dbcsr_amd/csrc/mm_numeric_f64_small.h) against the CPU oracle -- every cube from 5 to 8, rectangular triplets, mixes of sizes inside a launch (with
blocks of 1 ... 4 among them), tails, product lists longer than the 64 records a wave fetches at once, C blocks without products, alpha / beta,
retain_sparsity with in-place accumulation, transposes, a filtered multiply, both prefetch depths.  Index bit-exact, flop equal, values 1e-10 relative."""
import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, rel_err, to_dev

pytestmark = pytest.mark.gpu

# (M, N, K, sparsity A, B, C, mix m, mix n, mix k)
CASES = {
    "5cube": (5 * 40, 5 * 38, 5 * 42, 0.6, 0.6, 0.7, [1, 5], [1, 5], [1, 5]),
    "6cube": (6 * 40, 6 * 38, 6 * 42, 0.6, 0.6, 0.7, [1, 6], [1, 6], [1, 6]),
    "7cube_tails": (7 * 30 + 3, 7 * 31 + 5, 7 * 29 + 2, 0.6, 0.6, 0.7, [1, 7], [1, 7], [1, 7]),
    "8cube": (8 * 40, 8 * 38, 8 * 42, 0.6, 0.6, 0.7, [1, 8], [1, 8], [1, 8]),
    "8cube_tails": (8 * 30 + 5, 8 * 31 + 1, 8 * 29 + 7, 0.5, 0.5, 0.7, [1, 8], [1, 8], [1, 8]),
    "5x8x6": (5 * 40, 8 * 30, 6 * 35, 0.6, 0.6, 0.7, [1, 5], [1, 8], [1, 6]),
    "8x5x7": (8 * 30, 5 * 40, 7 * 33, 0.6, 0.6, 0.7, [1, 8], [1, 5], [1, 7]),
    "7x7_k3": (7 * 30, 7 * 32, 3 * 70, 0.6, 0.6, 0.7, [1, 7], [1, 7], [1, 3]),          # k <= 4: one MFMA per product
    "3x8_k8": (3 * 60, 8 * 30, 8 * 30, 0.6, 0.6, 0.7, [1, 3], [1, 8], [1, 8]),          # C blocks within 4 rows, not within 4 x 4
    "8x2_k5": (8 * 30, 2 * 90, 5 * 44, 0.6, 0.6, 0.7, [1, 8], [1, 2], [1, 5]),
    "mix_1_to_8": (230, 240, 250, 0.6, 0.6, 0.7, [1, 5, 1, 8, 1, 1, 1, 3, 2, 7], [1, 6, 1, 2, 1, 8, 1, 4], [1, 8, 1, 5, 1, 1, 1, 7, 1, 4]),
    "mix_5_8": (5 * 20 + 8 * 20, 5 * 21 + 8 * 19, 5 * 18 + 8 * 22, 0.6, 0.6, 0.7, [1, 5, 1, 8], [1, 8, 1, 5], [1, 5, 1, 8]),
    "long_lists": (8 * 12, 8 * 12, 6 * 260, 0.55, 0.55, 0.5, [1, 8], [1, 8], [1, 6]),    # ~50 products per C block, some lists beyond 64 records
    "very_long_lists": (5 * 6, 7 * 6, 8 * 400, 0.3, 0.3, 0.5, [1, 5], [1, 7], [1, 8]),   # ~200 products per C block: four batches of records
    "sparse_lists": (8 * 60, 8 * 60, 8 * 60, 0.93, 0.93, 0.8, [1, 8], [1, 8], [1, 8]),   # C blocks with 0 .. 2 products
    "one_block": (8, 7, 6, 0.0, 0.0, 0.0, [1, 8], [1, 7], [1, 6]),
}
ENV = ("DBCSR_AMD_MM_KERNEL", "DBCSR_AMD_MM_HOT", "DBCSR_AMD_MM_TINY", "DBCSR_AMD_MM_SYMBOLIC", "DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_WG_WAVES",
       "DBCSR_AMD_MM_SMALL", "DBCSR_AMD_MM_KCHUNKS", "DBCSR_AMD_MM_WORK")


def check(out, ref, tol=1e-10):
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert rel_err(out.data, ref.data) <= tol


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("alpha,beta,depth", [(0.7, 1.3, None), (1.0, 0.0, "4")])
def test_small_block_kernel_matches_oracle(monkeypatch, name, alpha, beta, depth):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    if depth:
        monkeypatch.setenv("DBCSR_AMD_MM_SMALL", depth)
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*CASES[name])
    ref, info = O.multiply("N", "N", alpha, A, B, beta, Cm)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", alpha, dA, dB, beta, dC, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f64_small<%s>" % (depth or "2"), eng.last_kernel()
    assert flop[0] == info["flop"]
    check(dev_to_bcsr(dC), ref)


@pytest.mark.parametrize("name", ["7cube_tails", "mix_1_to_8", "very_long_lists", "sparse_lists", "one_block"])
@pytest.mark.parametrize("depth,work", [("2", "0"), ("3", None), ("6", "0"), ("8", None)])
def test_small_block_kernel_depths_and_the_start_without_launch_records(monkeypatch, name, depth, work):
    """the other prefetch depths, and DBCSR_AMD_MM_WORK=0: the wave starts from order[] -> descs[] -> entries[] instead of the launch-order records"""
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DBCSR_AMD_MM_SMALL", depth)
    if work:
        monkeypatch.setenv("DBCSR_AMD_MM_WORK", work)
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*CASES[name])
    ref, info = O.multiply("N", "N", -1.5, A, B, 0.25, Cm)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", -1.5, dA, dB, 0.25, dC, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f64_small<%s>" % depth, eng.last_kernel()
    assert flop[0] == info["flop"]
    check(dev_to_bcsr(dC), ref)


@pytest.mark.parametrize("name", ["5cube", "8cube_tails", "mix_1_to_8", "long_lists", "sparse_lists"])
def test_small_block_kernel_retain_and_in_place(monkeypatch, name):
    """retain_sparsity, then a second product accumulated in place (C blocks without products in the call stay untouched: skip_empty)"""
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*CASES[name])
    ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm, retain_sparsity=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f64_small<2>"
    check(dev_to_bcsr(dC), ref)
    ref2, _ = O.multiply("N", "N", -0.5, A, B, 1.0, ref, retain_sparsity=True)
    dbcsr_multiply("N", "N", -0.5, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
    torch.cuda.synchronize()
    check(dev_to_bcsr(dC), ref2)


@pytest.mark.parametrize("ta,tb", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
def test_small_block_kernel_transposes_and_filter(monkeypatch, ta, tb):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    sm, sn, sk = O.make_block_sizes(5 * 30 + 3, [1, 5]), O.make_block_sizes(8 * 25 + 2, [1, 8]), O.make_block_sizes(7 * 28 + 4, [1, 7])
    c0 = O.RANDMAT_SEED_INIT
    Cm = O.make_random_matrix(sm, sn, 0.6, c0 + 1)
    A = O.make_random_matrix(sk, sm, 0.5, c0 + 2) if ta == "T" else O.make_random_matrix(sm, sk, 0.5, c0 + 2)
    B = O.make_random_matrix(sn, sk, 0.5, c0 + 3) if tb == "T" else O.make_random_matrix(sk, sn, 0.5, c0 + 3)
    for eps in (0.0, 100.0):
        ref, info = O.multiply(ta, tb, 1.5, A, B, 0.5, Cm, filter_eps=eps)
        if eps:
            full, finfo = O.multiply(ta, tb, 1.5, A, B, 0.5, Cm)
            assert info["flop"] < finfo["flop"] and ref.nblks < full.nblks, "the filter case does not filter"
        eng = MultiplyEngine()
        dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
        flop = [0]
        dbcsr_multiply(ta, tb, 1.5, dA, dB, 0.5, dC, filter_eps=eps or None, flop=flop, engine=eng)
        torch.cuda.synchronize()
        assert eng.last_kernel() == "mm_numeric_f64_small<2>", eng.last_kernel()
        assert flop[0] == info["flop"]
        out = dev_to_bcsr(dC)
        assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i)
        assert rel_err(out.data, ref.data) <= 1e-10


def test_small_block_kernel_agrees_with_the_staged_kernel(monkeypatch):
    """the same products in the same order with the same instruction through DBCSR_AMD_MM_SMALL=0 (the LDS-staged kernels of the sizes up to 32):
    the results differ by the rounding of the alpha / beta epilogue at most"""
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    A, B, Cm = O.perf_case(*CASES["mix_5_8"])
    outs = []
    for sw in (None, "0"):
        if sw:
            monkeypatch.setenv("DBCSR_AMD_MM_SMALL", sw)
        eng = MultiplyEngine()
        dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
        dbcsr_multiply("N", "N", 0.7, dA, dB, 1.3, dC, engine=eng)
        torch.cuda.synchronize()
        assert eng.last_kernel().startswith("mm_numeric_f64_small<2>" if sw is None else "mm_numeric_f64_"), eng.last_kernel()
        assert sw is None or "small" not in eng.last_kernel()
        outs.append(dev_to_bcsr(dC))
    assert np.array_equal(outs[0].col_i, outs[1].col_i) and rel_err(outs[0].data, outs[1].data) <= 1e-14


def test_blocks_within_4x4_keep_the_packed_kernel_and_9_keeps_the_exact_size_kernel(monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    for case, expect in (((4 * 40, 4 * 40, 8 * 20, 0.6, 0.6, 0.7, [1, 4], [1, 4], [1, 8]), "mm_numeric_f64_tiny"),
                         ((9 * 20, 8 * 20, 8 * 20, 0.6, 0.6, 0.7, [1, 9], [1, 8], [1, 8]), "mm_numeric_f64_"),
                         ((8 * 20, 8 * 20, 9 * 20, 0.6, 0.6, 0.7, [1, 8], [1, 8], [1, 9]), "mm_numeric_f64_")):
        A, B, Cm = O.perf_case(*case)
        ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
        eng = MultiplyEngine()
        dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
        dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
        torch.cuda.synchronize()
        assert eng.last_kernel().startswith(expect) and "small" not in eng.last_kernel(), eng.last_kernel()
        check(dev_to_bcsr(dC), ref)
