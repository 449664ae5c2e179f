"""GPU paths against outputs of the REAL reference library (tests/golden/ref_dump.json: true dbcsr_multiply on the
BLAS path, built unchanged with tools/build_dbcsr_host.py): the dbcsr_multiply mirror (symmetric / antisymmetric operands
included: they are passed as stored, one triangle, and desymmetrized on the device) and the one-call native dbcsr_amd_multiply.  Block index identical, flop identical, values within 1e-10 relative (north star); single-precision cases (config 5's data type) within 5e-6 of the largest element."""
import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import dbcsr_multiply
from tests import ref_dump_util as R
from tests.gpu_util import dev_to_bcsr, to_dev

pytestmark = pytest.mark.gpu


def compare(out, flop, ref):
    assert np.array_equal(out.row_p, ref.row_p), "row_p differs from the reference"
    assert np.array_equal(out.col_i, ref.col_i), "col_i differs from the reference"
    assert flop == ref.flop
    if ref.data is not None and ref.nblks:
        scale = max(np.max(np.abs(ref.data)), 1e-300)
        assert out.data.size == ref.data.size
        # single precision: reference and device both sum in float, in different orders
        tol = 5e-6 if R.np_dtype(ref.params) == np.float32 else 1e-10
        assert np.max(np.abs(out.data.astype(np.float64) - ref.data)) <= tol * scale


@pytest.mark.parametrize("name", R.names())
def test_mirror_matches_reference_dump(name):
    ref = R.RefResult(name)
    p = ref.params
    A, B, Cm = R.oracle_inputs(p)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dC.symmetry = p["symm_c"]   # a product matrix with symmetry comes (and goes back) as its stored triangle
    for which, (stored, symm) in R.stored_operands(p).items():  # symmetric operands go in as the reference stores them: one triangle
        d = to_dev(stored)
        d.symmetry = symm
        if which == "A":
            dA = d
        else:
            dB = d
    flop = [0]
    lim = [v or None for v in p["limits"]]
    dbcsr_multiply(p["transa"], p["transb"], p["alpha"], dA, dB, p["beta"], dC, first_row=lim[0], last_row=lim[1], first_column=lim[2],
                   last_column=lim[3], first_k=lim[4], last_k=lim[5], retain_sparsity=p["retain"],
                   filter_eps=p["filter_eps"] if p["filter_eps"] >= 0 else None, flop=flop)
    torch.cuda.synchronize()
    compare(dev_to_bcsr(dC), flop[0], ref)


@pytest.mark.parametrize("name", R.names(R.nonsymmetric))
def test_native_call_matches_reference_dump(name):
    from tests.test_gpu_native_multiply import native_multiply
    ref = R.RefResult(name)
    p = ref.params
    A, B, Cm = R.oracle_inputs(p)
    got, flop = native_multiply(p["transa"], p["transb"], p["alpha"], A, B, p["beta"], Cm, limits=p["limits"] if any(p["limits"]) else None,
                                retain=p["retain"], eps=max(p["filter_eps"], 0.0), dtype=R.np_dtype(p))
    compare(got, flop, ref)


@pytest.mark.parametrize("name", R.names(lambda p: p["symm_c"] != "N" and p["symm_a"] == "N" and p["symm_b"] == "N"))
def test_native_symmetric_product_matches_reference_dump(name):
    """dbcsr_amd_multiply_symmetric_c: canonical form, masked multiply, back to the stored triangle, in one native call."""
    import ctypes as C

    from dbcsr_amd import lib as L
    from dbcsr_amd.multiply import MultiplyEngine
    from oracle import oracle as O
    from tests.test_gpu_native_multiply import fetch
    ref = R.RefResult(name)
    p = ref.params
    A, B, Cm = R.oracle_inputs(p)
    E = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    a, b, c = dA.desc(), dB.desc(), dC.desc()
    out, flop = L.BcsrDesc(), C.c_int64(0)
    rc = E.L.dbcsr_amd_multiply_symmetric_c(E.h, p["transa"].encode(), p["transb"].encode(), L.dbcsr_type_real_8, float(p["alpha"]), C.byref(a),
                                            C.byref(b), float(p["beta"]), C.byref(c), 1 if p["symm_c"] == "A" else 0, 1 if p["retain"] else 0,
                                            float(max(p["filter_eps"], 0.0)), C.byref(out), C.byref(flop), None)
    assert rc == 0
    torch.cuda.synchronize()
    nbr, nblks = out.nblkrows, int(out.nblks)
    row_p = fetch(E.L, out.row_p, nbr + 1, np.int32)
    col_i = fetch(E.L, out.col_i, nblks, np.int32)
    blk_p = fetch(E.L, out.blk_p, nblks, np.int64)
    rows = np.repeat(np.arange(nbr), np.diff(row_p))
    nze = int((Cm.row_sizes[rows].astype(np.int64) * Cm.col_sizes[col_i]).sum()) if nblks else 0
    data = fetch(E.L, out.data, nze, np.float64)
    assert E.L.dbcsr_amd_bcsr_release(C.byref(out)) == 0
    compare(O.Bcsr(Cm.row_sizes, Cm.col_sizes, row_p, col_i, blk_p, data), flop.value, ref)


@pytest.mark.parametrize("symbolic", ["word", "grid", "rows"])
@pytest.mark.parametrize("name", ["symm_c_S_NT", "symm_c_A_NT", "symm_c_S_retain", "symm_abc_S"])
def test_symmetric_product_with_every_symbolic_kernel_family(name, symbolic, monkeypatch):
    """the canonical-form mask sits in the C bitmap: the three families of counting / filling kernels must all honour it"""
    from dbcsr_amd.multiply import MultiplyEngine
    monkeypatch.setenv("DBCSR_AMD_MM_SYMBOLIC", symbolic)
    E = MultiplyEngine()
    ref = R.RefResult(name)
    p = ref.params
    A, B, Cm = R.oracle_inputs(p)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    dC.symmetry = p["symm_c"]
    for which, (stored, symm) in R.stored_operands(p).items():
        d = to_dev(stored)
        d.symmetry = symm
        if which == "A":
            dA = d
        else:
            dB = d
    flop = [0]
    dbcsr_multiply(p["transa"], p["transb"], p["alpha"], dA, dB, p["beta"], dC, retain_sparsity=p["retain"], flop=flop, engine=E)
    torch.cuda.synchronize()
    compare(dev_to_bcsr(dC), flop[0], ref)
