"""The tile dataflow of the fp64 engine (dbcsr_amd/csrc/mm_tile.h: XCD-wide C tiles in registers, one k-sorted product list per
3 x 3 sub-tile, team-wide k window) forced on cases far below its automatic threshold and compared with the CPU oracle: index
bit-exact, values 1e-10.  The cases cover C blocks that are new / present in C_in, inner blocks of another size (the remainder
pass), block rows and columns of another size (left to the exact-size kernel), sub-tiles with fewer than nine C blocks (sparse C,
matrix edges), several super-tiles per XCD, retain_sparsity, and every throttle setting (the window only affects speed)."""
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
MANY = (23 * 170 + 16, 23 * 150 + 16, 23 * 160 + 16, 0.9, 0.9, 0.9, [1, 23], [1, 23], [1, 23])       # 16 super-tiles: two per XCD
TAILS = (23 * 40 + 16 + 7, 23 * 44 + 9, 23 * 42 + 16 + 5, 0.7, 0.7, 0.8, [20, 23, 1, 16, 1, 7], [22, 23, 1, 9], [21, 23, 1, 16, 1, 5])

ENV_KEYS = ("DBCSR_AMD_MM_TILE", "DBCSR_AMD_MM_TILE_SHAPE", "DBCSR_AMD_MM_TILE_WINDOW", "DBCSR_AMD_MM_TILE_RDV", "DBCSR_AMD_MM_TILE_PUB", "DBCSR_AMD_MM_TILE_PREFETCH", "DBCSR_AMD_MM_KERNEL", "DBCSR_AMD_MM_HOT", "DBCSR_AMD_MM_CLASSES",
            "DBCSR_AMD_MM_DBG", "DBCSR_AMD_MM_WG_WAVES")


def run(monkeypatch, env, case, alpha=0.7, beta=1.3, retain=False):
    for k in ENV_KEYS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DBCSR_AMD_LAB", "1")   # the build with the experimental dataflows (dbcsr_amd/csrc/Makefile)
    monkeypatch.setenv("DBCSR_AMD_MM_TILE", "2")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*case)
    ref, info = O.multiply("N", "N", alpha, A, B, beta, Cm, retain_sparsity=retain)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", alpha, dA, dB, beta, dC, retain_sparsity=retain, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("mm_numeric_f64_tile<23,23,23>"), eng.last_kernel()
    gave_up, mismatches = eng.tile_stats()
    assert mismatches == 0, "sub-tile product lists disagree with the per-block product counts"
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert flop[0] == info["flop"]
    assert rel_err(out.data, ref.data) <= 1e-10
    return gave_up


# shape 0: 3 x 3 C blocks per wave, two waves per SIMD, two-slot ring; shape 1: 4 x 3, one wave per SIMD, four-slot ring (mm_tile.h)
SHAPES = ["0", "1"]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("case", [H2O, DENSE, SPARSE_C, MANY, TAILS], ids=["h2o", "dense", "sparse_c", "many_tiles", "tails"])
def test_tile_kernel_matches_oracle(monkeypatch, case, shape):
    run(monkeypatch, {"DBCSR_AMD_MM_TILE_SHAPE": shape}, case)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("case", [H2O, MANY], ids=["h2o", "many_tiles"])
def test_tile_kernel_retain_sparsity(monkeypatch, case, shape):
    run(monkeypatch, {"DBCSR_AMD_MM_TILE_SHAPE": shape}, case, alpha=1.0, beta=1.0, retain=True)


@pytest.mark.parametrize("window", ["0", "1", "8", "64", "100000"])
def test_tile_kernel_deep_ring_any_window(monkeypatch, window):
    """shape 1: results never depend on the team protocol, and short lists (fewer products than ring slots) drain correctly"""
    run(monkeypatch, {"DBCSR_AMD_MM_TILE_WINDOW": window, "DBCSR_AMD_MM_TILE_SHAPE": "1"}, MANY)
    run(monkeypatch, {"DBCSR_AMD_MM_TILE_WINDOW": window, "DBCSR_AMD_MM_TILE_SHAPE": "1"}, SPARSE_C)


@pytest.mark.parametrize("window", ["0", "1", "8", "64", "100000"])
def test_tile_kernel_any_window(monkeypatch, window):
    # the k window of the team is a speed knob: a window of one inner block serialises the team, none lets every wave run free
    gave_up = run(monkeypatch, {"DBCSR_AMD_MM_TILE_WINDOW": window}, MANY)
    assert gave_up == 0


@pytest.mark.parametrize("env", [{"DBCSR_AMD_MM_TILE_PUB": "0"}, {"DBCSR_AMD_MM_TILE_PUB": "1"}, {"DBCSR_AMD_MM_TILE_PREFETCH": "1"},
                                 {"DBCSR_AMD_MM_TILE_PREFETCH": "1", "DBCSR_AMD_MM_TILE_WINDOW": "16"}],
                         ids=lambda e: "-".join("%s=%s" % (k[18:], v) for k, v in e.items()))
def test_tile_kernel_store_policy_and_prefetch(monkeypatch, env):
    # progress stores written through or left in the XCD's L2, operand blocks pulled into L2 one product earlier: speed knobs only
    assert run(monkeypatch, env, MANY) == 0
    run(monkeypatch, env, H2O)


def test_tile_kernel_unpaired_fragment_reads(monkeypatch):
    run(monkeypatch, {"DBCSR_AMD_MM_TILE_RDV": "1"}, H2O)


def test_tile_kernel_beta_zero_and_new_c(monkeypatch):
    # beta = 0 without retain_sparsity: the multiply empties C first (src/mm/dbcsr_mm.F:865-870), every C block is new (cin_off = -1)
    case = (23 * 30 + 16, 23 * 30 + 16, 23 * 30 + 16, 0.8, 0.8, 0.7, [1, 23], [1, 23], [1, 23])
    run(monkeypatch, {}, case, alpha=1.0, beta=0.0)


def test_tile_kernel_not_chosen_for_small_or_filtered(monkeypatch):
    for k in ENV_KEYS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DBCSR_AMD_LAB", "1")
    monkeypatch.setenv("DBCSR_AMD_MM_TILE", "1")   # automatic: the case is far below the threshold
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(*H2O)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("mm_numeric_f64_hot<23,23,23>"), eng.last_kernel()
    assert eng.tile_stats() is None
