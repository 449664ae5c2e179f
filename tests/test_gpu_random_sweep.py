"""Randomised parity sweep: 160 seeded multiplies with random shapes, block-size mixes (single size, mixed, tails, tiny blocks), fills,
transposes, alpha / beta (including 0 and 1), retain_sparsity, filter_eps and product-matrix symmetry through the dbcsr_multiply mirror
against the CPU oracle -- index bit-exact, flop identical, values 1e-10 (fp64) / 2e-5 (fp32).  The hand-picked cases of the other
files pin known corners; this one walks the combinations nobody thought of (empty rows, C blocks without products, single-block
matrices, lists longer and shorter than the kernels' windows) through whatever kernel the engine picks."""
import os

import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, to_dev

pytestmark = pytest.mark.gpu

MIXES = [[1, 23], [1, 13, 1, 23, 1, 32], [1, 5], [1, 4], [1, 32], [2, 7, 1, 32], [1, 1, 1, 3], [1, 16, 1, 8], [3, 9, 1, 2], [1, 29, 1, 31]]


# a soak with cases no earlier walk has seen: DBCSR_AMD_SWEEP_OFFSET shifts every generator seed
SWEEP_OFFSET = int(os.environ.get("DBCSR_AMD_SWEEP_OFFSET", "0"))


def make_case(seed):
    rng = np.random.default_rng(seed + SWEEP_OFFSET)
    mix_m, mix_n, mix_k = (MIXES[int(rng.integers(len(MIXES)))] for _ in range(3))
    symm_c = "N" if rng.random() < 0.8 else ("S" if rng.random() < 0.7 else "A")
    if symm_c != "N":
        mix_n = mix_m
    M = int(rng.integers(1, 420))
    N = M if symm_c != "N" else int(rng.integers(1, 420))
    K = int(rng.integers(1, 420))
    sp = tuple(float(x) for x in rng.choice([0.0, 0.3, 0.6, 0.8, 0.95, 0.999], size=3))
    ta, tb = (str(x) for x in rng.choice(["N", "T"], size=2))
    alpha = float(rng.choice([1.0, -0.5, 0.0, 2.25]))
    beta = float(rng.choice([1.0, 0.0, -1.5, 0.5]))
    retain = bool(rng.random() < 0.2)
    eps = float(rng.choice([0.0, 0.0, 0.0, 3.0, 25.0]))
    dtype = np.float32 if (symm_c == "N" and rng.random() < 0.2) else np.float64
    return dict(M=M, N=N, K=K, sp=sp, mix_m=mix_m, mix_n=mix_n, mix_k=mix_k, ta=ta, tb=tb, alpha=alpha, beta=beta, retain=retain, eps=eps,
                symm_c=symm_c, dtype=dtype)


FORCED = [{"DBCSR_AMD_MM_CLASSES": "2"}, {"DBCSR_AMD_MM_SYMBOLIC": "rows"}, {"DBCSR_AMD_MM_CLASSES": "2", "DBCSR_AMD_MM_SYMBOLIC": "rows"},
          {"DBCSR_AMD_MM_WG_WAVES": "4", "DBCSR_AMD_MM_CLASSES": "2"}, {"DBCSR_AMD_MM_SYMBOLIC": "word"}, {"DBCSR_AMD_MM_HOT": "0"}]


@pytest.mark.parametrize("seed", range(int(os.environ.get("DBCSR_AMD_SWEEP_FORCED", "60"))))   # (a longer walk: set the variable)
def test_random_multiply_forced_paths(seed, monkeypatch):
    """the same sweep with the run-time compiled class kernels, the product-driven / per-word symbolic kernels, four waves per workgroup
    or the generic kernels forced (they engage by themselves only at sizes the oracle needs minutes for)"""
    for k in ("DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_SYMBOLIC", "DBCSR_AMD_MM_WG_WAVES", "DBCSR_AMD_MM_HOT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in FORCED[seed % len(FORCED)].items():
        monkeypatch.setenv(k, v)
    run_case(make_case(5000 + seed))


@pytest.mark.parametrize("seed", range(int(os.environ.get("DBCSR_AMD_SWEEP_PLAIN", "160"))))
def test_random_multiply_matches_oracle(seed):
    run_case(make_case(1000 + seed))


# round 5: the same walk over matrices with blocks of 33 ... 80 (the workgroup-per-C-block kernel mm_numeric_f64_big, mixed with small blocks,
# tails, empty rows), a seed range of its own so that the cases of the walks above stay what they were
BIG_MIXES = [[1, 45], [1, 72], [1, 67, 1, 5], [1, 40, 1, 23], [1, 80], [2, 33, 1, 13], [1, 64], [1, 55, 1, 3], [1, 78, 1, 32], [1, 23]]


# round 6: C blocks of 33 ... 48 in both dimensions (the one-wave kernel mm_numeric_f64_mid and its second launch for the blocks of another size)
MID_MIXES = [[1, 33], [1, 36], [1, 40], [1, 37, 1, 34], [3, 35, 1, 8], [1, 39, 1, 40], [1, 38], [1, 44], [1, 48], [1, 41, 1, 47], [4, 45, 1, 30]]


def make_big_case(seed, mixes=BIG_MIXES, k_mixes=BIG_MIXES):
    rng = np.random.default_rng(seed + SWEEP_OFFSET)
    mix_m, mix_n = (mixes[int(rng.integers(len(mixes)))] for _ in range(2))
    mix_k = k_mixes[int(rng.integers(len(k_mixes)))]
    symm_c = "N" if rng.random() < 0.85 else ("S" if rng.random() < 0.7 else "A")
    if symm_c != "N":
        mix_n = mix_m
    M = int(rng.integers(1, 600))
    N = M if symm_c != "N" else int(rng.integers(1, 600))
    K = int(rng.integers(1, 600))
    sp = tuple(float(x) for x in rng.choice([0.0, 0.3, 0.6, 0.8, 0.95], size=3))
    ta, tb = (str(x) for x in rng.choice(["N", "T"], size=2))
    alpha = float(rng.choice([1.0, -0.5, 0.0, 2.25]))
    beta = float(rng.choice([1.0, 0.0, -1.5, 0.5]))
    retain = bool(rng.random() < 0.2)
    eps = float(rng.choice([0.0, 0.0, 0.0, 30.0, 200.0]))
    return dict(M=M, N=N, K=K, sp=sp, mix_m=mix_m, mix_n=mix_n, mix_k=mix_k, ta=ta, tb=tb, alpha=alpha, beta=beta, retain=retain, eps=eps,
                symm_c=symm_c, dtype=np.float64)


@pytest.mark.parametrize("seed", range(int(os.environ.get("DBCSR_AMD_SWEEP_BIG", "60"))))
def test_random_multiply_large_blocks(seed):
    run_case(make_big_case(9000 + seed))


@pytest.mark.parametrize("seed", range(int(os.environ.get("DBCSR_AMD_SWEEP_MID", "40"))))
def test_random_multiply_blocks_of_33_to_40(seed):
    run_case(make_big_case(12000 + seed, MID_MIXES, BIG_MIXES + MID_MIXES))


def build_case(c):
    """(A, B, C_in, reference result, reference info) of a case on the host -- everything the oracle contributes"""
    sm, sn, sk = O.make_block_sizes(c["M"], c["mix_m"]), O.make_block_sizes(c["N"], c["mix_n"]), O.make_block_sizes(c["K"], c["mix_k"])
    c0 = O.RANDMAT_SEED_INIT
    dt = c["dtype"]
    Cm = (O.make_random_matrix(sm, sn, c["sp"][2], c0 + 1, dt) if c["symm_c"] == "N"
          else O.make_random_matrix_symmetric(sm, c["sp"][2], c0 + 1, c["symm_c"]))
    A = O.make_random_matrix(sk, sm, c["sp"][0], c0 + 2, dt) if c["ta"] == "T" else O.make_random_matrix(sm, sk, c["sp"][0], c0 + 2, dt)
    B = O.make_random_matrix(sn, sk, c["sp"][1], c0 + 3, dt) if c["tb"] == "T" else O.make_random_matrix(sk, sn, c["sp"][1], c0 + 3, dt)
    wide = lambda X: O.Bcsr(X.row_sizes, X.col_sizes, X.row_p, X.col_i, X.blk_p, X.data.astype(np.float64))
    ref, info = O.multiply(c["ta"], c["tb"], c["alpha"], wide(A), wide(B), c["beta"], wide(Cm), retain_sparsity=c["retain"], filter_eps=c["eps"],
                           c_symmetry=None if c["symm_c"] == "N" else c["symm_c"])
    return A, B, Cm, ref, info


def device_case(c, A, B, Cm):
    """(result on the host, flop) of the case through the dbcsr_multiply mirror on a fresh engine"""
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dC.symmetry = c["symm_c"]
    flop = [0]
    dbcsr_multiply(c["ta"], c["tb"], c["alpha"], dA, dB, c["beta"], dC, retain_sparsity=c["retain"], filter_eps=c["eps"] or None, flop=flop,
                   engine=MultiplyEngine())
    torch.cuda.synchronize()
    return dev_to_bcsr(dC), flop[0]


def compare_case(c, out, flop, ref, info):
    """None when the device result equals the reference's by the bar of this file, else a dict that says what differs"""
    if not (np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i)):
        return {"what": "index", "nblks_device": int(out.col_i.size), "nblks_oracle": int(ref.col_i.size)}
    if flop != info["flop"]:
        return {"what": "flop", "device": int(flop), "oracle": int(info["flop"])}
    if ref.data.size:
        scale = max(float(np.max(np.abs(ref.data))), 1e-300)
        tol = 1e-10 if c["dtype"] == np.float64 else 2e-5
        if out.data.size != ref.data.size:
            return {"what": "data size", "device": int(out.data.size), "oracle": int(ref.data.size)}
        err = np.abs(out.data.astype(np.float64) - ref.data)
        worst = int(np.argmax(err))
        if not float(err[worst]) <= tol * scale:    # (a NaN fails too)
            return {"what": "values", "worst_element": worst, "device": float(out.data[worst]), "oracle": float(ref.data[worst]),
                    "err_over_scale": float(err[worst]) / scale, "tol": tol, "count_above_tol": int(np.sum(~(err <= tol * scale)))}
    return None


def run_case(c):
    A, B, Cm, ref, info = build_case(c)
    out, flop = device_case(c, A, B, Cm)
    bad = compare_case(c, out, flop, ref, info)
    assert bad is None, (bad, c)
