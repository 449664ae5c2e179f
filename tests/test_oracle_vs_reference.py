"""Pins the oracle's stack-level restatement on the REFERENCE'S OWN host functions
(src/acc/libsmm_acc/libsmm_acc_benchmark.cpp: matInit, stackInit, stackCalc, stackTransp, checkSum,
checkSumTransp), compiled from the reference sources where they lie by oracle/build_ref.sh into
oracle/_ref/ref_stack_driver.  Skipped when that binary has not been built."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_stack_driver")
pytestmark = pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/ref_stack_driver not built (needs /root/reference)")


@pytest.mark.parametrize("m,n,k", [(23, 23, 23), (5, 4, 3), (13, 32, 7), (4, 4, 4)])
def test_stack_validator_matches_reference(m, n, k):
    na, nb, nc, ns = 100, 100, 10, 100  # the reference's `test` mode sizes (libsmm_acc_benchmark.cpp:45-52)
    ref = json.loads(subprocess.check_output([BIN] + [str(x) for x in (m, n, k, na, nb, nc, ns)]))
    a, b = O.mat_init(na, m, k, 42), O.mat_init(nb, k, n, 24)
    assert a[0] == ref["a0"] and b[-1] == ref["b_last"]
    stack = O.stack_init(ns, nc, na, nb, m, n, k, rseed=1)
    assert list(stack) == ref["stack"]
    c = np.zeros(nc * m * n)
    O.stack_calc(stack, c, a, b, m, n, k, b_transposed=True)  # stackCalc indexes B as n x k (:134-137)
    assert np.array_equal(c, np.asarray(ref["c"]))
    assert O.lib().orc_check_sum(c, c.size) == ref["checksum"]
    # transposition: the reference keeps the source and writes a transposed copy; the oracle works in place
    at = a.copy()
    O.transpose((np.arange(na, dtype=np.int32) * m * k), at, m, k)
    size, nsamp = m * k, (m * k) // 3
    step = size // nsamp if nsamp > 0 else size
    cs = sum(float(at[s * size + idx]) for s in range(na) for idx in range(s % step, size, step))
    assert cs == ref["checksum_transp"]  # checkSumTransp's sampling (:172-191)
