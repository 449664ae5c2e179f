"""dbcsr_amd_multiply: the whole dbcsr_multiply orchestration (op(), limits, retain_sparsity, filter_eps) as ONE native call
of the C-ABI, results owned by hipMalloc -- checked against the oracle (structure bit-exact, values 1e-10 / 1e-5 for fp32)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from dbcsr_amd import lib as L
from dbcsr_amd.multiply import MultiplyEngine
from oracle import oracle as O
from tests.gpu_util import to_dev
from tests.test_oracle_limits import CASES, limit_case_matrices

pytestmark = pytest.mark.gpu


def fetch(lib, ptr, count, dtype):
    out = np.empty(count, dtype)
    if count:
        assert lib.c_dbcsr_acc_memcpy_d2h(C.c_void_p(ptr), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes), None) == 0
        assert lib.c_dbcsr_acc_device_synchronize() == 0
    return out


def native_multiply(transa, transb, alpha, A, B, beta, Cm, limits=None, retain=False, eps=0.0, dtype=np.float64):
    E = MultiplyEngine()
    lib = E.L
    cast = lambda M: O.Bcsr(M.row_sizes, M.col_sizes, M.row_p, M.col_i, M.blk_p, M.data.astype(dtype))
    dA, dB, dC = to_dev(cast(A)), to_dev(cast(B)), to_dev(cast(Cm))
    a, b, c = dA.desc(), dB.desc(), dC.desc()
    out = L.BcsrDesc()
    flop = C.c_int64(0)
    lim = (C.c_int64 * 6)(*limits) if limits is not None else None
    code = L.dbcsr_type_real_8 if dtype == np.float64 else L.dbcsr_type_real_4
    rc = lib.dbcsr_amd_multiply(E.h, transa.encode(), transb.encode(), code, float(alpha), C.byref(a), C.byref(b), float(beta), C.byref(c),
                                lim, 1 if retain else 0, float(eps), C.byref(out), C.byref(flop), None)
    assert rc == 0
    torch.cuda.synchronize()
    nbr = out.nblkrows
    row_p = fetch(lib, out.row_p, nbr + 1, np.int32)
    nblks = int(out.nblks)
    assert row_p[-1] == nblks
    col_i = fetch(lib, out.col_i, nblks, np.int32)
    blk_p = fetch(lib, out.blk_p, nblks, np.int64)
    rows = np.repeat(np.arange(nbr), np.diff(row_p))
    nze = int((Cm.row_sizes[rows].astype(np.int64) * Cm.col_sizes[col_i]).sum()) if nblks else 0
    data = fetch(lib, out.data, nze, dtype)
    assert lib.dbcsr_amd_bcsr_release(C.byref(out)) == 0
    return O.Bcsr(Cm.row_sizes, Cm.col_sizes, row_p, col_i, blk_p, data), flop.value


def same(got, ref, tol):
    assert np.array_equal(got.row_p, ref.row_p) and np.array_equal(got.col_i, ref.col_i) and np.array_equal(got.blk_p, ref.blk_p)
    assert np.all(np.abs(got.data.astype(np.float64) - ref.data) <= tol * np.maximum(np.abs(ref.data), 1.0))


@pytest.mark.parametrize("trans", ["NN", "TN", "NT", "TT"])
@pytest.mark.parametrize("alpha,beta,retain", [(1.0, 1.0, False), (-0.5, 2.0, False), (2.0, 0.0, True)])
def test_native_multiply_matches_oracle(trans, alpha, beta, retain):
    A, B, Cm = O.perf_case(230, 260, 200, 0.5, 0.6, 0.7, [1, 13, 1, 5], [1, 23, 1, 4], [1, 7, 1, 32], trans[0], trans[1])
    ref, info = O.multiply(trans[0], trans[1], alpha, A, B, beta, Cm, retain_sparsity=retain)
    got, flop = native_multiply(trans[0], trans[1], alpha, A, B, beta, Cm, retain=retain)
    same(got, ref, 1e-10)
    assert flop == info["flop"]


@pytest.mark.parametrize("case", [c for c in CASES if c[0] in ("BETA", "LIMITS_MIX_3", "LIMITS_MIX_7", "LIMITS_ROW_3", "CUT_NEW")], ids=lambda c: c[0])
def test_native_multiply_limits(case):
    A, B, Cm = limit_case_matrices(case)
    _, _, _, retain, alpha, beta, _, _, _, lim = case
    ref, info = O.multiply_limits("N", "N", alpha, A, B, beta, Cm, lim, retain_sparsity=retain)
    got, flop = native_multiply("N", "N", alpha, A, B, beta, Cm, limits=lim, retain=retain)
    same(got, ref, 1e-10)
    assert flop == info["flop"]


@pytest.mark.parametrize("eps", [2.0, 40.0])
def test_native_multiply_filter_eps(eps):
    A, B, Cm = O.perf_case(300, 260, 280, 0.5, 0.5, 0.5, [1, 5, 1, 13], [1, 7, 1, 9], [1, 4, 1, 23])
    ref, info = O.multiply("N", "N", 1.0, A, B, 1.0, Cm, filter_eps=eps)
    got, flop = native_multiply("N", "N", 1.0, A, B, 1.0, Cm, eps=eps)
    same(got, ref, 1e-10)


def test_native_multiply_fp32_and_errors():
    A, B, Cm = O.perf_case(320, 320, 320, 0.6, 0.6, 0.6, [1, 32], [1, 32], [1, 32])
    ref, _ = O.multiply("N", "N", 0.5, A, B, 2.0, Cm)
    got, _ = native_multiply("N", "N", 0.5, A, B, 2.0, Cm, dtype=np.float32)
    assert np.array_equal(got.row_p, ref.row_p) and np.array_equal(got.col_i, ref.col_i)
    assert float(np.max(np.abs(got.data - ref.data))) <= 1e-5 * float(np.max(np.abs(ref.data)))
    E = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    a, b, c = dA.desc(), dB.desc(), dC.desc()
    out = L.BcsrDesc()
    assert E.L.dbcsr_amd_multiply(E.h, b"X", b"N", L.dbcsr_type_real_8, 1.0, C.byref(a), C.byref(b), 1.0, C.byref(c), None, 0, 0.0, C.byref(out),
                                  None, None) != 0
    lim = (C.c_int64 * 6)(10, 5, 0, 0, 0, 0)
    assert E.L.dbcsr_amd_multiply(E.h, b"N", b"N", L.dbcsr_type_real_8, 1.0, C.byref(a), C.byref(b), 1.0, C.byref(c), lim, 0, 0.0, C.byref(out),
                                  None, None) != 0
    assert E.L.dbcsr_amd_multiply(E.h, b"N", b"N", 7, 1.0, C.byref(a), C.byref(b), 1.0, C.byref(c), None, 0, 0.0, C.byref(out), None, None) == -10


C_EXAMPLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "multiply_example")


@pytest.mark.skipif(not os.path.exists(C_EXAMPLE), reason="tests/c/multiply_example not built (__graft_entry__.build())")
def test_plain_c_host_example():
    r = subprocess.run([C_EXAMPLE], capture_output=True, timeout=120)
    assert r.returncode == 0, (r.stdout.decode()[-500:], r.stderr.decode()[-1500:])
    assert b"multiply_example:" in r.stdout


@pytest.mark.parametrize("npass", [2, 5])
def test_native_multiply_k_passes(npass, monkeypatch):
    monkeypatch.setenv("DBCSR_AMD_MM_KCHUNKS", str(npass))
    A, B, Cm = O.perf_case(260, 240, 300, 0.5, 0.6, 0.7, [1, 13, 1, 5], [1, 23, 1, 4], [1, 7, 1, 32, 1, 9])
    ref, info = O.multiply("N", "N", -1.5, A, B, 0.5, Cm)
    got, flop = native_multiply("N", "N", -1.5, A, B, 0.5, Cm)
    same(got, ref, 1e-10)
    assert flop == info["flop"]
