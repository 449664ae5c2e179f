"""GPU parity of dbcsr_multiply's submatrix limits (first_row ... last_k): device crop / window scale / multiply against
the oracle (structure bit-exact, values 1e-10) and against the dense-window check of the reference's unit test."""
import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, to_dev
from tests.test_oracle_limits import CASES, dense_expected, limit_case_matrices, reference_criterion

pytestmark = pytest.mark.gpu


def close(x, ref, tol=1e-10):
    x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
    return x.shape == ref.shape and (x.size == 0 or bool(np.all(np.abs(x - ref) <= tol * np.maximum(np.abs(ref), 1.0))))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("trans", ["NN", "TN", "NT"])
def test_limits_match_oracle_and_dense_window(case, trans):
    transa, transb = trans[0], trans[1]
    if trans != "NN" and case[0] not in ("BETA", "LIMITS_MIX_6", "LIMITS_MIX_7", "CUT_NEW"):
        pytest.skip("transposes on a subset")
    A, B, Cm = limit_case_matrices(case, transa, transb)
    _, _, _, retain, alpha, beta, _, _, _, lim = case
    ref, info = O.multiply_limits(transa, transb, alpha, A, B, beta, Cm, lim, retain_sparsity=retain)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply(transa, transb, alpha, dA, dB, beta, dC, first_row=lim[0], last_row=lim[1], first_column=lim[2],
                   last_column=lim[3], first_k=lim[4], last_k=lim[5], retain_sparsity=retain, flop=flop)
    torch.cuda.synchronize()
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert flop[0] == info["flop"]
    assert close(out.data, ref.data)
    Ad, Bd, Cd, E = dense_expected(case, A, B, Cm, transa, transb)
    assert reference_criterion(Ad, Bd, Cd, E, out.to_dense(), lim[3] - lim[2] + 1) <= 10.0


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_crop_and_scale_window_bit_exact(dtype):
    M = O.make_random_matrix(O.make_block_sizes(61, [1, 5, 2, 3]), O.make_block_sizes(47, [1, 4, 1, 7]), 0.4, 4711, dtype)
    E = MultiplyEngine()
    for rb, cb in [((7, 40), (5, 33)), (None, (10, 12)), ((0, 60), None), ((30, 30), (20, 20)), ((58, 60), (0, 3))]:
        ref = O.crop(M, rb, cb)
        out = dev_to_bcsr(E.cropped(to_dev(M), rb, cb))
        assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
        assert np.array_equal(out.data, ref.data)
        sref = O.scale_window(M, dtype(-2.5), rb, cb)
        sout = dev_to_bcsr(E.scaled_window(to_dev(M), -2.5, rb, cb))
        assert np.array_equal(sout.data, sref.data)


def test_invalid_limits_raise():
    A, B, Cm = limit_case_matrices(CASES[2])
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    with pytest.raises(ValueError):
        dbcsr_multiply("N", "N", 1.0, dA, dB, 0.0, dC, first_row=10, last_row=5)
    with pytest.raises(ValueError):
        dbcsr_multiply("N", "N", 1.0, dA, dB, 0.0, dC, last_column=51)
