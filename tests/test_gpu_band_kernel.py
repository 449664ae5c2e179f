"""The band dataflow of the fp64 engine (dbcsr_amd/csrc/mm_band.h: CU-wide C tiles of 24 x 3 blocks, the B operand shared by the
eight waves of a workgroup in an LDS ring with reference counts) forced on cases far below its automatic threshold and compared with
the CPU oracle: index bit-exact, values 1e-10.  The cases cover C blocks that are new / present in C_in, inner blocks of another size
(the remainder pass), block rows and columns of another size (left to the exact-size kernel), sub-tiles with fewer than nine C blocks
and tiles with fewer than eight sub-tiles (sparse C, matrix edges), workgroups with several tiles and with none, every ring depth
(the depth only affects speed: a shallow ring makes the waves wait for each other all the time) and both cache policies of the B
copies."""
import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, rel_err, to_dev

pytestmark = pytest.mark.gpu

H2O = (23 * 20 + 16, 23 * 18 + 16, 23 * 22 + 16, 0.6, 0.6, 0.7, [1, 23], [1, 23], [1, 23])
DENSE = (23 * 31, 23 * 29, 23 * 33, 0.0, 0.0, 0.0, [1, 23], [1, 23], [1, 23])                        # every product, no tails
SPARSE_C = (23 * 60 + 16, 23 * 64 + 16, 23 * 62 + 16, 0.93, 0.93, 0.95, [1, 23], [1, 23], [1, 23])   # most sub-tiles incomplete
MANY = (23 * 170 + 16, 23 * 150 + 16, 23 * 160 + 16, 0.9, 0.9, 0.9, [1, 23], [1, 23], [1, 23])       # 8 bands x 54 column triples: two tiles per workgroup
TAILS = (23 * 40 + 16 + 7, 23 * 44 + 9, 23 * 42 + 16 + 5, 0.7, 0.7, 0.8, [20, 23, 1, 16, 1, 7], [22, 23, 1, 9], [21, 23, 1, 16, 1, 5])
WIDE = (23 * 30, 23 * 40, 23 * 700, 0.85, 0.85, 0.9, [1, 23], [1, 23], [1, 23])                      # 2 bands x 234 column triples: long sweeps per workgroup

ENV_KEYS = ("DBCSR_AMD_MM_BAND", "DBCSR_AMD_MM_BAND_SHAPE", "DBCSR_AMD_MM_BAND_WINDOW", "DBCSR_AMD_MM_BAND_DEPTH", "DBCSR_AMD_MM_BAND_BPOL", "DBCSR_AMD_MM_BAND_KNOBS", "DBCSR_AMD_MM_TILE", "DBCSR_AMD_MM_KERNEL",
            "DBCSR_AMD_MM_HOT", "DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_DBG", "DBCSR_AMD_MM_WG_WAVES", "DBCSR_AMD_MM_HOT_PERSISTENT")


def run(monkeypatch, env, case, alpha=0.7, beta=1.3, reps=1):
    for k in ENV_KEYS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DBCSR_AMD_LAB", "1")   # the build with the experimental dataflows (dbcsr_amd/csrc/Makefile)
    monkeypatch.setenv("DBCSR_AMD_MM_BAND", "2")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*case)
    ref, info = O.multiply("N", "N", alpha, A, B, beta, Cm)
    dA, dB = to_dev(A), to_dev(B)
    for _ in range(reps):   # the second multiply reuses the plan (lists and all)
        dC = to_dev(Cm)
        flop = [0]
        dbcsr_multiply("N", "N", alpha, dA, dB, beta, dC, flop=flop, engine=eng)
        torch.cuda.synchronize()
        assert eng.last_kernel().startswith("mm_numeric_f64_band<23,23,23>"), eng.last_kernel()
        gave_up, mismatches = eng.band_stats()
        assert mismatches == 0, "(tile, wave) product lists disagree with the per-block product counts"
        assert gave_up == 0, "a wait of the ring protocol gave up"
        out = dev_to_bcsr(dC)
        assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
        assert flop[0] == info["flop"]
        assert rel_err(out.data, ref.data) <= 1e-10


# shape 0: 8 waves per workgroup, 3 x 3 C blocks per wave, two A slots; shape 1: 16 waves, 2 x 2, one A slot (mm_band.h)
SHAPES = ["0", "1"]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("case", [H2O, DENSE, SPARSE_C, MANY, TAILS, WIDE], ids=["h2o", "dense", "sparse_c", "many_tiles", "tails", "wide"])
def test_band_kernel_matches_oracle(monkeypatch, case, shape):
    run(monkeypatch, {"DBCSR_AMD_MM_BAND_SHAPE": shape}, case)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("depth", ["12", "16", "20", "22"])
def test_band_kernel_any_ring_depth(monkeypatch, depth, shape):
    run(monkeypatch, {"DBCSR_AMD_MM_BAND_DEPTH": depth, "DBCSR_AMD_MM_BAND_SHAPE": shape}, MANY)
    run(monkeypatch, {"DBCSR_AMD_MM_BAND_DEPTH": depth, "DBCSR_AMD_MM_BAND_SHAPE": shape}, DENSE)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("window", ["0", "1", "8", "64", "100000"])
def test_band_kernel_any_window(monkeypatch, window, shape):
    # the k window of an XCD's waves is a speed knob: a window of one inner block serialises them, none lets every wave run free
    run(monkeypatch, {"DBCSR_AMD_MM_BAND_WINDOW": window, "DBCSR_AMD_MM_BAND_SHAPE": shape}, MANY)
    run(monkeypatch, {"DBCSR_AMD_MM_BAND_WINDOW": window, "DBCSR_AMD_MM_BAND_SHAPE": shape}, SPARSE_C)


@pytest.mark.parametrize("shape", SHAPES)
def test_band_kernel_streaming_b_copies_and_timing_knob(monkeypatch, shape):
    run(monkeypatch, {"DBCSR_AMD_MM_BAND_BPOL": "1", "DBCSR_AMD_MM_BAND_SHAPE": shape}, MANY)
    run(monkeypatch, {"DBCSR_AMD_MM_BAND_KNOBS": "1", "DBCSR_AMD_MM_BAND_SHAPE": shape}, H2O)


@pytest.mark.parametrize("shape", SHAPES)
def test_band_kernel_plan_reuse(monkeypatch, shape):
    run(monkeypatch, {"DBCSR_AMD_MM_BAND_SHAPE": shape}, MANY, reps=3)


def test_band_kernel_beta_zero_and_new_c(monkeypatch):
    # beta = 0 without retain_sparsity: the multiply empties C first (src/mm/dbcsr_mm.F:865-870), every C block is new (cin_off = -1)
    case = (23 * 30 + 16, 23 * 30 + 16, 23 * 30 + 16, 0.8, 0.8, 0.7, [1, 23], [1, 23], [1, 23])
    run(monkeypatch, {}, case, alpha=1.0, beta=0.0)


def test_band_kernel_not_chosen_for_small_retained_or_filtered(monkeypatch):
    for k in ENV_KEYS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DBCSR_AMD_LAB", "1")
    monkeypatch.setenv("DBCSR_AMD_MM_BAND", "1")   # automatic: the case is far below the threshold
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*H2O)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("mm_numeric_f64_hot<23,23,23>"), eng.last_kernel()
    assert eng.band_stats() is None
    monkeypatch.setenv("DBCSR_AMD_MM_BAND", "2")   # forced, but retain_sparsity takes the pattern from C_in: not a band case
    eng = MultiplyEngine()
    ref, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm, retain_sparsity=True)
    dC = to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("mm_numeric_f64_hot<23,23,23>"), eng.last_kernel()
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.col_i, ref.col_i) and rel_err(out.data, ref.data) <= 1e-10
