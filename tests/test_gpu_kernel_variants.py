"""Every block-product kernel variant forced through the oracle: the automatic choice of
dbcsr_amd_mm_numeric picks one kernel per launch from the sizes of the case, so a parity suite that only uses the defaults
covers the other variants by accident.  Here the switches DBCSR_AMD_MM_KERNEL / _HOT / _TINY (read when an engine is
created) force each variant on cases it can run, the name of the kernel that ran is asserted, and the result is compared
with the CPU oracle (index bit-exact, values 1e-10 fp64 / 1e-5 fp32)."""
import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, rel_err, to_dev

pytestmark = pytest.mark.gpu

MIXED = (300, 280, 260, 0.5, 0.5, 0.6, [1, 13, 1, 23, 1, 32, 1, 7], [1, 23, 1, 5, 1, 32], [1, 13, 1, 32, 1, 9])
H2O = (23 * 20 + 16, 23 * 18 + 16, 23 * 22 + 16, 0.6, 0.6, 0.7, [1, 23], [1, 23], [1, 23])
TINY = (240, 240, 240, 0.7, 0.7, 0.7, [1, 4], [1, 4, 1, 3], [1, 4, 1, 2])
TINY_K = (230, 250, 420, 0.6, 0.7, 0.7, [1, 4, 1, 2], [1, 3, 1, 1, 1, 4], [1, 7, 1, 4, 1, 13])  # C blocks of at most 4 x 4, k extents above 4
TINY_FEW = (400, 400, 400, 0.9, 0.9, 0.93, [1, 4], [1, 4], [1, 4])  # lists of zero to a few products: mostly idle places in a chunk
CONFIG3 = (68 * 9 + 24, 68 * 8 + 24, 68 * 10 + 24, 0.8, 0.8, 0.8, [1, 13, 1, 23, 1, 32], [1, 13, 1, 23, 1, 32], [1, 13, 1, 23, 1, 32])
# config 3's regime at an oracle-friendly size: {13, 23, 32} blocks + tail 24, 200 block rows, fill 13.6 % -> 3.7 products per C block
CONFIG3_37 = (68 * 66 + 24, 68 * 66 + 24, 68 * 66 + 24, 0.864, 0.864, 0.864, [1, 13, 1, 23, 1, 32], [1, 13, 1, 23, 1, 32], [1, 13, 1, 23, 1, 32])
POW2 = (16 * 20 + 9, 32 * 10 + 5, 16 * 18 + 3, 0.6, 0.6, 0.6, [1, 16, 1, 32], [1, 32, 1, 16, 1, 8], [1, 16, 1, 32, 1, 24])  # padded LDS pitches
SPARSE = (23 * 120 + 16, 23 * 110 + 16, 23 * 130 + 16, 0.97, 0.97, 0.98, [1, 23], [1, 23], [1, 23])  # 0.1 products per candidate
BIG = (300, 270, 280, 0.5, 0.5, 0.5, [1, 45, 1, 13], [1, 67, 1, 5], [1, 40, 1, 23])

# (environment, case, expected kernel-name prefix)
VARIANTS = [
    ({}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    # experiment variants of the exact-size kernel: run-time ablation switches (1, all off), unpaired fragment reads (2), touches of the
    # neighbouring A blocks (3, 4)
    ({"DBCSR_AMD_MM_HOT_VARIANT": "1"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    ({"DBCSR_AMD_MM_HOT_VARIANT": "2"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    ({"DBCSR_AMD_MM_HOT_VARIANT": "3"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    ({"DBCSR_AMD_MM_HOT_VARIANT": "4"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    # raised issue priority for a product's MFMA burst (5), the same with unpaired fragment reads (6)
    ({"DBCSR_AMD_MM_HOT_VARIANT": "5"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    ({"DBCSR_AMD_MM_HOT_VARIANT": "6"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    # the same kernel as persistent waves with a work counter per XCD
    ({"DBCSR_AMD_MM_HOT_PERSISTENT": "1"}, H2O, "mm_numeric_f64_hot_persistent<23,23,23>"),
    ({"DBCSR_AMD_MM_HOT": "0", "DBCSR_AMD_MM_KERNEL": "lds1"}, H2O, "mm_numeric_f64_lds<3>"),
    ({"DBCSR_AMD_MM_KERNEL": "pipe"}, H2O, "mm_numeric_f64_pipe<3>"),
    ({"DBCSR_AMD_MM_KERNEL": "dma2"}, H2O, "mm_numeric_f64_dma<23,23,23,2>"),
    ({"DBCSR_AMD_MM_KERNEL": "dma3"}, H2O, "mm_numeric_f64_dma<23,23,23,3>"),
    ({"DBCSR_AMD_MM_KERNEL": "direct"}, H2O, "mm_numeric_f64"),
    ({"DBCSR_AMD_MM_KERNEL": "lds1"}, MIXED, "mm_numeric_f64_lds<4>"),
    ({"DBCSR_AMD_MM_KERNEL": "pipe"}, MIXED, "mm_numeric_f64_pipe<4>"),
    ({"DBCSR_AMD_MM_KERNEL": "pipe", "DBCSR_AMD_MM_PIPE_G": "3"}, MIXED, "mm_numeric_f64_pipe<4>"),
    ({"DBCSR_AMD_MM_KERNEL": "direct"}, MIXED, "mm_numeric_f64"),
    ({}, TINY, "mm_numeric_f64_tiny"),
    ({}, TINY_K, "mm_numeric_f64_tiny"),
    ({}, TINY_FEW, "mm_numeric_f64_tiny"),
    ({"DBCSR_AMD_MM_TINY": "0", "DBCSR_AMD_MM_KERNEL": "lds1"}, TINY, "mm_numeric_f64_lds<1>"),
    ({"DBCSR_AMD_MM_TINY": "0", "DBCSR_AMD_MM_KERNEL": "pipe"}, TINY, "mm_numeric_f64_pipe<1>"),
    ({}, BIG, "mm_numeric_f64"),
    ({"DBCSR_AMD_MM_SYMBOLIC": "word"}, MIXED, "mm_numeric_f64"),
    # product-driven symbolic kernels (automatic for a sparse C with many block rows: BASELINE config 4)
    ({"DBCSR_AMD_MM_SYMBOLIC": "rows"}, MIXED, "mm_numeric_f64"),
    ({"DBCSR_AMD_MM_SYMBOLIC": "rows"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    ({"DBCSR_AMD_MM_SYMBOLIC": "rows", "DBCSR_AMD_MM_CLASSES": "2"}, CONFIG3_37, "mm_numeric_f64_class["),
    ({"DBCSR_AMD_MM_SYMBOLIC": "rows"}, SPARSE, "mm_numeric_f64"),
    # (m, n) classes with run-time compiled exact-size kernels (forced: the cases are far below the automatic threshold)
    ({"DBCSR_AMD_MM_CLASSES": "2"}, MIXED, "mm_numeric_f64_class["),
    ({"DBCSR_AMD_MM_CLASSES": "2"}, H2O, "mm_numeric_f64_class["),
    ({"DBCSR_AMD_MM_CLASSES": "2"}, CONFIG3, "mm_numeric_f64_class["),
    ({"DBCSR_AMD_MM_CLASSES": "2"}, POW2, "mm_numeric_f64_class["),
    # (round 6: the classes (32, 32), (32, 23), (23, 32) take the one-wave slab kernels mm_numeric_f64_mid<8,8> / <8,6> / <6,8>; DBCSR_AMD_MM_MID=0: all
    # nine through the class kernels; DBCSR_AMD_MM_MID=3: only (32, 32) through the slab kernel)
    ({"DBCSR_AMD_MM_CLASSES": "2"}, CONFIG3_37, "mm_numeric_f64_class[6 jit + 3 slab + 1 generic"),
    ({"DBCSR_AMD_MM_CLASSES": "2", "DBCSR_AMD_MM_MID": "0"}, CONFIG3_37, "mm_numeric_f64_class[9 jit + 1 generic"),
    ({"DBCSR_AMD_MM_CLASSES": "2", "DBCSR_AMD_MM_MID": "3"}, CONFIG3_37, "mm_numeric_f64_class[8 jit + 1 slab + 1 generic"),
    ({"DBCSR_AMD_MM_CLASSES": "0"}, CONFIG3_37, "mm_numeric_f64_pipe<4>"),
    # the class kernels with a wave walking 8 / 4 consecutive C blocks of its class (product pipeline across block boundaries)
    ({"DBCSR_AMD_MM_CLASSES": "2", "DBCSR_AMD_MM_CLASS_G": "8"}, CONFIG3_37, "mm_numeric_f64_class[9 jit + 1 generic"),
    ({"DBCSR_AMD_MM_CLASSES": "2", "DBCSR_AMD_MM_CLASS_G": "8"}, MIXED, "mm_numeric_f64_class["),
    ({"DBCSR_AMD_MM_CLASSES": "2", "DBCSR_AMD_MM_CLASS_G": "4"}, POW2, "mm_numeric_f64_class["),
    ({"DBCSR_AMD_MM_CLASSES": "2", "DBCSR_AMD_MM_CLASS_G": "8"}, H2O, "mm_numeric_f64_class["),   # what the automatic choice runs below the class threshold
    # four / two / one waves per workgroup (the default goes by the mean product-list length)
    ({"DBCSR_AMD_MM_WG_WAVES": "4"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    ({"DBCSR_AMD_MM_WG_WAVES": "2"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    ({"DBCSR_AMD_MM_WG_WAVES": "1"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    ({"DBCSR_AMD_MM_WG_WAVES": "1", "DBCSR_AMD_MM_KERNEL": "lds1", "DBCSR_AMD_MM_HOT": "0"}, MIXED, "mm_numeric_f64_lds"),
    ({"DBCSR_AMD_MM_WG_WAVES": "4", "DBCSR_AMD_MM_KERNEL": "lds1", "DBCSR_AMD_MM_HOT": "0"}, MIXED, "mm_numeric_f64_lds"),
    ({"DBCSR_AMD_MM_WG_WAVES": "4", "DBCSR_AMD_MM_CLASSES": "2"}, CONFIG3_37, "mm_numeric_f64_class[6 jit + 3 slab + 1 generic"),
    ({"DBCSR_AMD_MM_WG_WAVES": "2", "DBCSR_AMD_MM_CLASSES": "2"}, MIXED, "mm_numeric_f64_class["),
    # without the launch-order work records (the default has them): order[] -> descs[] -> entries[]
    ({"DBCSR_AMD_MM_WORK": "0"}, H2O, "mm_numeric_f64_hot<23,23,23>"),
    ({"DBCSR_AMD_MM_WORK": "0", "DBCSR_AMD_MM_CLASSES": "2"}, CONFIG3_37, "mm_numeric_f64_class[6 jit + 3 slab + 1 generic"),
    ({"DBCSR_AMD_MM_WORK": "0", "DBCSR_AMD_MM_CLASSES": "2"}, MIXED, "mm_numeric_f64_class["),
    # the class launches of one multiply spread over several streams
    ({"DBCSR_AMD_MM_CLASS_STREAMS": "3", "DBCSR_AMD_MM_CLASSES": "2"}, CONFIG3_37, "mm_numeric_f64_class[6 jit + 3 slab + 1 generic"),
    ({"DBCSR_AMD_MM_CLASS_STREAMS": "4", "DBCSR_AMD_MM_CLASSES": "2"}, MIXED, "mm_numeric_f64_class["),
]


LAB_SWITCHES = ("DBCSR_AMD_MM_HOT_VARIANT", "DBCSR_AMD_MM_HOT_PERSISTENT", "DBCSR_AMD_MM_HOT_XCDS", "DBCSR_AMD_MM_CLASS_G", "DBCSR_AMD_MM_CLASS_STREAMS",
                "DBCSR_AMD_MM_DBG", "DBCSR_AMD_MM_LDS_PAD", "DBCSR_AMD_MM_ROW_GROUP")


def run_case(monkeypatch, env, case, dtype, tol, expect, alpha=0.7, beta=1.3, retain=False, in_place_twice=False):
    for k in ("DBCSR_AMD_MM_KERNEL", "DBCSR_AMD_MM_HOT", "DBCSR_AMD_MM_TINY", "DBCSR_AMD_MM_PIPE_G", "DBCSR_AMD_MM_SYMBOLIC", "DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_CLASS_G", "DBCSR_AMD_MM_WORK", "DBCSR_AMD_MM_WG_WAVES", "DBCSR_AMD_MM_CLASS_STREAMS", "DBCSR_AMD_MM_HOT_VARIANT", "DBCSR_AMD_MM_HOT_PERSISTENT", "DBCSR_AMD_MM_HOT_XCDS", "DBCSR_AMD_MM_F32_DIRECT", "DBCSR_AMD_MM_BIG", "DBCSR_AMD_MM_MID"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    # the switches of the experimental variants exist in the lab build only (dbcsr_amd/csrc/Makefile); everything else runs on the shipping one
    lab = any(k in LAB_SWITCHES for k in env) or env.get("DBCSR_AMD_MM_KERNEL", "").startswith("dma")
    eng = MultiplyEngine(lab=lab)  # reads the switches now
    A, B, Cm = O.perf_case(*case)
    ref, info = O.multiply("N", "N", alpha, A, B, beta, Cm, retain_sparsity=retain)
    cast = lambda M: O.Bcsr(M.row_sizes, M.col_sizes, M.row_p, M.col_i, M.blk_p, M.data.astype(dtype))
    dA, dB, dC = to_dev(cast(A)), to_dev(cast(B)), to_dev(cast(Cm))
    flop = [0]
    dbcsr_multiply("N", "N", alpha, dA, dB, beta, dC, retain_sparsity=retain, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().replace(" ", "").startswith(expect.replace(" ", "")), (eng.last_kernel(), expect)
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert flop[0] == info["flop"]
    assert rel_err(out.data, ref.data) <= tol
    if in_place_twice:  # accumulate a second product into the result (retain_sparsity, beta = 1: the Cannon tick path)
        ref2, _ = O.multiply("N", "N", alpha, A, B, 1.0, ref, retain_sparsity=True)
        dbcsr_multiply("N", "N", alpha, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
        torch.cuda.synchronize()
        out2 = dev_to_bcsr(dC)
        assert np.array_equal(out2.col_i, ref2.col_i) and rel_err(out2.data, ref2.data) <= tol


@pytest.mark.parametrize("env,case,expect", VARIANTS, ids=lambda v: "-".join("%s=%s" % (k[13:], x) for k, x in v.items()) if isinstance(v, dict) else None)
def test_fp64_variant_matches_oracle(monkeypatch, env, case, expect):
    run_case(monkeypatch, env, case, np.float64, 1e-10, expect)


@pytest.mark.parametrize("env,case,expect", [v for v in VARIANTS if v[1] in (H2O, MIXED)][:9] + [v for v in VARIANTS if "DBCSR_AMD_MM_CLASSES" in v[0] or v[0].get("DBCSR_AMD_MM_SYMBOLIC") == "rows"],
                         ids=lambda v: "-".join("%s=%s" % (k[13:], x) for k, x in v.items()) if isinstance(v, dict) else None)
def test_fp64_variant_retain_and_in_place(monkeypatch, env, case, expect):
    run_case(monkeypatch, env, case, np.float64, 1e-10, expect, alpha=1.0, beta=1.0, retain=True, in_place_twice=True)


F32 = (32 * 12, 32 * 11, 32 * 13, 0.6, 0.6, 0.6, [1, 32], [1, 32], [1, 32])
F32_TAILS = (32 * 24 + 20, 32 * 22 + 7, 32 * 26 + 12, 0.6, 0.6, 0.6, [1, 32], [1, 32], [1, 32])
F32_16 = (16 * 30 + 5, 16 * 28 + 9, 16 * 33 + 4, 0.6, 0.6, 0.6, [1, 16], [1, 16], [1, 16])
F32_24 = (24 * 20 + 13, 24 * 18, 24 * 22 + 8, 0.6, 0.6, 0.6, [1, 24], [1, 24], [1, 24])
F32_23 = (23 * 20 + 16, 23 * 18 + 16, 23 * 22 + 16, 0.6, 0.6, 0.7, [1, 23], [1, 23], [1, 23])
F32_MIXED = (300, 280, 260, 0.5, 0.5, 0.6, [1, 13, 1, 32, 1, 7], [1, 23, 1, 32], [1, 13, 1, 32, 1, 9])


@pytest.mark.parametrize("env,case,expect", [
    ({}, F32, "mm_numeric_f32_direct<32,32,32>"),
    ({}, F32_TAILS, "mm_numeric_f32_direct<32,32,32>"),   # tail blocks: C blocks of other sizes and products with another inner dimension
    ({}, F32_16, "mm_numeric_f32_direct<16,16,16>"),
    ({}, F32_24, "mm_numeric_f32_direct<24,24,24>"),
    ({}, F32_23, "mm_numeric_f32_hot<23,23,23>"),         # k not a multiple of 8: the kernel that stages both operands
    ({"DBCSR_AMD_MM_F32_DIRECT": "0"}, F32, "mm_numeric_f32_hot<32,32,32>"),
    ({"DBCSR_AMD_MM_F32_DIRECT": "0"}, F32_TAILS, "mm_numeric_f32_hot<32,32,32>"),
    ({"DBCSR_AMD_MM_HOT": "0"}, F32, "mm_numeric_f32_lds"),
    ({"DBCSR_AMD_MM_KERNEL": "direct"}, F32, "mm_numeric_f32"),
    ({}, F32_MIXED, "mm_numeric_f32_lds"),
    ({"DBCSR_AMD_MM_KERNEL": "direct"}, F32_MIXED, "mm_numeric_f32"),
    ({}, BIG, "mm_numeric_f32"),
    ({"DBCSR_AMD_MM_CLASSES": "2"}, F32_MIXED, "mm_numeric_f32_lds[per class"),
    ({"DBCSR_AMD_MM_WG_WAVES": "4"}, F32_TAILS, "mm_numeric_f32_direct<32,32,32>"),
    ({"DBCSR_AMD_MM_WG_WAVES": "1"}, F32_TAILS, "mm_numeric_f32_direct<32,32,32>"),
    ({"DBCSR_AMD_MM_WG_WAVES": "2"}, F32_16, "mm_numeric_f32_direct<16,16,16>"),
    ({"DBCSR_AMD_MM_WG_WAVES": "2"}, F32_MIXED, "mm_numeric_f32_lds"),
    ({"DBCSR_AMD_MM_WG_WAVES": "4", "DBCSR_AMD_MM_CLASSES": "2"}, F32_MIXED, "mm_numeric_f32_lds[per class"),
], ids=lambda v: "-".join("%s=%s" % (k[13:], x) for k, x in v.items()) if isinstance(v, dict) else None)
def test_fp32_variant_matches_oracle(monkeypatch, env, case, expect):
    run_case(monkeypatch, env, case, np.float32, 2e-5, expect, alpha=1.0, beta=1.0)
