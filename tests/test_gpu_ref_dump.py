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
