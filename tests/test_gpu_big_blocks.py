"""Blocks of 33 ... 80 (libsmm_acc's range: max_kernel_dim = 80, src/core/dbcsr_config.F:185; libsmm_acc.cpp:324-339) -- round 6: 33 ... 40 in
both dimensions through the one-wave kernel mm_numeric_f64_mid (see CASES), the rest through the
workgroup-per-C-block kernel mm_numeric_f64_big (one 2 x 2 arrangement of waves per block, operand slabs of 16 inner indices shared
through LDS) against the CPU oracle: every sub-block shape TM x TN in {2 .. 5}^2 that the host can choose, inner dimensions with every
remainder modulo 4 and 16 (slab and k-step tails), blocks smaller than the launch's largest in the same launch, C blocks without
products, alpha / beta, retain_sparsity with in-place accumulation, transposes, a filtered multiply.  Index bit-exact, flop equal,
values 1e-10 relative -- the bar of the whole suite."""
import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, rel_err, to_dev

pytestmark = pytest.mark.gpu

# (M, N, K, sparsity A, B, C, mix m, mix n, mix k), expected kernel
CASES = {
    "72cube": ((72 * 5, 72 * 4, 72 * 6, 0.4, 0.4, 0.5, [1, 72], [1, 72], [1, 72]), "mm_numeric_f64_big<5,5>"),
    "80cube_tails": ((80 * 3 + 33, 80 * 3 + 7, 80 * 4 + 50, 0.3, 0.3, 0.5, [1, 80], [1, 80], [1, 80]), "mm_numeric_f64_big<5,5>"),
    "64cube": ((64 * 5, 64 * 5, 64 * 5, 0.4, 0.4, 0.5, [1, 64], [1, 64], [1, 64]), "mm_numeric_f64_big<4,4>"),
    "40cube": ((40 * 8, 40 * 7, 40 * 9, 0.5, 0.5, 0.5, [1, 40], [1, 40], [1, 40]), "mm_numeric_f64_mid<10,10>"),
    "33cube": ((33 * 8, 33 * 9, 33 * 7, 0.5, 0.5, 0.5, [1, 33], [1, 33], [1, 33]), "mm_numeric_f64_mid<9,9>"),
    "55cube": ((55 * 6, 55 * 5, 55 * 7, 0.5, 0.5, 0.5, [1, 55], [1, 55], [1, 55]), "mm_numeric_f64_big<4,4>"),
    "45x67x78": ((45 * 7, 67 * 5, 78 * 5, 0.4, 0.4, 0.5, [1, 45], [1, 67], [1, 78]), "mm_numeric_f64_big<3,5>"),
    "78x45x67": ((78 * 4, 45 * 7, 67 * 5, 0.4, 0.4, 0.5, [1, 78], [1, 45], [1, 67]), "mm_numeric_f64_big<5,3>"),
    "23x23_k78": ((23 * 12, 23 * 11, 78 * 5, 0.4, 0.4, 0.5, [1, 23], [1, 23], [1, 78]), "mm_numeric_f64_big<2,2>"),    # small C blocks, long inner dimension
    "80x16_k37": ((80 * 4, 16 * 12, 37 * 9, 0.4, 0.4, 0.5, [1, 80], [1, 16], [1, 37]), "mm_numeric_f64_big<5,2>"),
    "13x72_k33": ((13 * 14, 72 * 4, 33 * 9, 0.4, 0.4, 0.5, [1, 13], [1, 72], [1, 33]), "mm_numeric_f64_big<2,5>"),
    "mixed_sizes": ((400, 390, 410, 0.5, 0.5, 0.6, [1, 45, 1, 13, 1, 72, 1, 5], [1, 67, 1, 5, 1, 33], [1, 40, 1, 23, 1, 3, 1, 61]), "mm_numeric_f64_big<5,5>"),
    "k_remainders": ((48 * 5, 56 * 5, 420, 0.4, 0.4, 0.5, [1, 48], [1, 56], [1, 17, 1, 18, 1, 19, 1, 33, 1, 34, 1, 35, 1, 49, 1, 1, 1, 64]), "mm_numeric_f64_big<3,4>"),
    "sparse_lists": ((72 * 9, 72 * 9, 72 * 9, 0.85, 0.85, 0.7, [1, 72], [1, 72], [1, 72]), "mm_numeric_f64_big<5,5>"),   # C blocks with 0 .. 2 products
    "one_block": ((72, 80, 33, 0.0, 0.0, 0.0, [1, 72], [1, 80], [1, 33]), "mm_numeric_f64_big<5,5>"),
    # round 6: blocks of 33 ... 40 in both dimensions -- one wave per C block in units of 4 x 4 (mm_numeric_f64_mid<rows / 4, columns / 4>): 9 x 9 with
    # both edges, 9 x 10 / 10 x 9 with one, 10 x 10 with none; the blocks of another size (tail row / column, mixes) through the second launch
    "34cube": ((34 * 8, 34 * 9, 34 * 7, 0.5, 0.5, 0.5, [1, 34], [1, 34], [1, 34]), "mm_numeric_f64_mid<9,9>"),
    "35x36x37": ((35 * 8, 36 * 8, 37 * 7, 0.5, 0.5, 0.5, [1, 35], [1, 36], [1, 37]), "mm_numeric_f64_mid<9,9>"),
    "37x33x36": ((37 * 8, 33 * 9, 36 * 7, 0.5, 0.5, 0.5, [1, 37], [1, 33], [1, 36]), "mm_numeric_f64_mid<10,9>"),
    "36x39_k80": ((36 * 8, 39 * 8, 80 * 4, 0.5, 0.5, 0.5, [1, 36], [1, 39], [1, 80]), "mm_numeric_f64_mid<9,10>"),
    "36cube_tails": ((36 * 6 + 20, 36 * 6 + 7, 36 * 6 + 30, 0.4, 0.4, 0.5, [1, 36], [1, 36], [1, 36]), "mm_numeric_f64_mid<9,9>"),
    "mix_33_to_40": ((36 * 8, 36 * 8, 300, 0.5, 0.5, 0.5, [1, 33, 1, 40, 1, 37, 1, 36], [1, 40, 1, 34, 1, 38], [1, 40, 1, 5, 1, 33, 1, 17]), "mm_numeric_f64_mid<10,10>"),
    "mostly_34_some_small": ((34 * 12 + 13, 34 * 12 + 40, 34 * 8, 0.5, 0.5, 0.5, [12, 34, 1, 13], [12, 34, 1, 40], [1, 34]), "mm_numeric_f64_mid<"),
    # no dominant size: the slab kernel only when ONE launch serves every block (round 6, session 47: 30 / 36 mixed lost 0.68 against the workgroup kernel)
    "mix_30_36": ((33 * 8, 33 * 8, 33 * 7, 0.5, 0.5, 0.5, [1, 30, 1, 36], [1, 36, 1, 30], [1, 30, 1, 36]), "mm_numeric_f64_big<3,3>"),
    "mix_33_36": ((35 * 8, 35 * 8, 35 * 7, 0.5, 0.5, 0.5, [1, 33, 1, 36], [1, 36, 1, 33], [1, 33, 1, 36]), "mm_numeric_f64_mid<9,9>"),
    "mix_23_40": ((32 * 8, 32 * 8, 32 * 7, 0.5, 0.5, 0.5, [1, 23, 1, 40], [1, 40, 1, 23], [1, 23, 1, 40]), "mm_numeric_f64_mid<10,10>"),
    # ... and 41 ... 48 (11 / 12 units): the largest shape is then <12,12>
    "44cube": ((44 * 7, 44 * 6, 44 * 8, 0.5, 0.5, 0.5, [1, 44], [1, 44], [1, 44]), "mm_numeric_f64_mid<11,11>"),
    "48cube_tails": ((48 * 6 + 20, 48 * 6 + 45, 48 * 6 + 30, 0.4, 0.4, 0.5, [1, 48], [1, 48], [1, 48]), "mm_numeric_f64_mid<12,12>"),
    "41x47_k33": ((41 * 7, 47 * 6, 33 * 9, 0.5, 0.5, 0.5, [1, 41], [1, 47], [1, 33]), "mm_numeric_f64_big<3,3>"),   # (both above 40, not multiples of 4: the workgroup kernel)
    "44x48_k33": ((44 * 7, 48 * 6, 33 * 9, 0.5, 0.5, 0.5, [1, 44], [1, 48], [1, 33]), "mm_numeric_f64_mid<11,12>"),
    "36x45_k80": ((36 * 8, 45 * 7, 80 * 4, 0.5, 0.5, 0.5, [1, 36], [1, 45], [1, 80]), "mm_numeric_f64_mid<9,12>"),
    "mix_30_to_48": ((400, 410, 300, 0.5, 0.5, 0.5, [1, 33, 1, 48, 1, 41, 1, 30], [1, 44, 1, 34, 1, 48], [1, 40, 1, 5, 1, 33, 1, 17]), "mm_numeric_f64_mid<12,12>"),
    # the workgroup kernel beside it: every parity of ceil(m / 8) tiles split between the two wave rows / columns
    "41x49x20": ((41 * 7, 49 * 6, 20 * 12, 0.5, 0.5, 0.5, [1, 41], [1, 49], [1, 20]), "mm_numeric_f64_big<3,4>"),
    "53x64x41": ((53 * 6, 64 * 5, 41 * 7, 0.5, 0.5, 0.5, [1, 53], [1, 64], [1, 41]), "mm_numeric_f64_big<4,4>"),
    "65x73x16": ((65 * 5, 73 * 4, 16 * 15, 0.4, 0.4, 0.5, [1, 65], [1, 73], [1, 16]), "mm_numeric_f64_big<5,5>"),
    "69x77x31": ((69 * 4, 77 * 4, 31 * 9, 0.4, 0.4, 0.5, [1, 69], [1, 77], [1, 31]), "mm_numeric_f64_big<5,5>"),
    "33x80x5": ((33 * 9, 80 * 4, 5 * 40, 0.4, 0.4, 0.5, [1, 33], [1, 80], [1, 5]), "mm_numeric_f64_big<3,5>"),
    "mixed_odd_sizes": ((420, 410, 400, 0.5, 0.5, 0.6, [1, 33, 1, 37, 1, 41, 1, 9, 1, 72, 1, 69], [1, 35, 1, 65, 1, 4, 1, 53, 1, 80], [1, 40, 1, 23, 1, 3, 1, 61]),
                        "mm_numeric_f64_big<5,5>"),
}
ENV = ("DBCSR_AMD_MM_KERNEL", "DBCSR_AMD_MM_HOT", "DBCSR_AMD_MM_TINY", "DBCSR_AMD_MM_SYMBOLIC", "DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_WG_WAVES",
       "DBCSR_AMD_MM_BIG", "DBCSR_AMD_MM_MID", "DBCSR_AMD_MM_KCHUNKS")


def check(out, ref, tol=1e-10):
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert rel_err(out.data, ref.data) <= tol


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("alpha,beta", [(0.7, 1.3), (1.0, 0.0)])
def test_big_block_kernel_matches_oracle(monkeypatch, name, alpha, beta):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    case, expect = CASES[name]
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*case)
    ref, info = O.multiply("N", "N", alpha, A, B, beta, Cm)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", alpha, dA, dB, beta, dC, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == expect or (expect.endswith("<") and eng.last_kernel().startswith(expect)), (eng.last_kernel(), expect)
    assert flop[0] == info["flop"]
    check(dev_to_bcsr(dC), ref)


@pytest.mark.parametrize("name", ["72cube", "45x67x78", "mixed_sizes", "sparse_lists", "36cube_tails", "mix_33_to_40", "48cube_tails", "mix_30_to_48"])
def test_big_block_kernel_retain_and_in_place(monkeypatch, name):
    """retain_sparsity, then a second product accumulated in place (C blocks without products in the call stay untouched: skip_empty)"""
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    case, expect = CASES[name]
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*case)
    ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm, retain_sparsity=True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == expect
    check(dev_to_bcsr(dC), ref)
    ref2, _ = O.multiply("N", "N", -0.5, A, B, 1.0, ref, retain_sparsity=True)
    dbcsr_multiply("N", "N", -0.5, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
    torch.cuda.synchronize()
    check(dev_to_bcsr(dC), ref2)


@pytest.mark.parametrize("ta,tb", [("T", "N"), ("N", "T"), ("T", "T")])
def test_big_block_kernel_transposes_and_filter(monkeypatch, ta, tb):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    sm, sn, sk = O.make_block_sizes(45 * 6, [1, 45]), O.make_block_sizes(67 * 4 + 20, [1, 67]), O.make_block_sizes(78 * 4 + 9, [1, 78])
    c0 = O.RANDMAT_SEED_INIT
    Cm = O.make_random_matrix(sm, sn, 0.5, c0 + 1)
    A = O.make_random_matrix(sk, sm, 0.4, c0 + 2) if ta == "T" else O.make_random_matrix(sm, sk, 0.4, c0 + 2)
    B = O.make_random_matrix(sn, sk, 0.4, c0 + 3) if tb == "T" else O.make_random_matrix(sk, sn, 0.4, c0 + 3)
    for eps in (0.0, 40.0):
        ref, info = O.multiply(ta, tb, 1.5, A, B, 0.5, Cm, filter_eps=eps)
        eng = MultiplyEngine()
        dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
        flop = [0]
        dbcsr_multiply(ta, tb, 1.5, dA, dB, 0.5, dC, filter_eps=eps or None, flop=flop, engine=eng)
        torch.cuda.synchronize()
        assert eng.last_kernel().startswith("mm_numeric_f64_big<"), eng.last_kernel()
        assert flop[0] == info["flop"]
        out = dev_to_bcsr(dC)
        assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i)
        assert rel_err(out.data, ref.data) <= 1e-10


def test_blocks_above_80_keep_the_plain_kernel(monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    A, B, Cm = O.perf_case(100 * 3, 90 * 3, 85 * 3, 0.3, 0.3, 0.5, [1, 100], [1, 90], [1, 85])
    ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
    eng = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f64"
    check(dev_to_bcsr(dC), ref)
    # and the switch that takes the blocks of 33 ... 80 back to it
    monkeypatch.setenv("DBCSR_AMD_MM_BIG", "0")
    case, _ = CASES["72cube"]
    A, B, Cm = O.perf_case(*case)
    ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
    eng = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "mm_numeric_f64"
    check(dev_to_bcsr(dC), ref)
