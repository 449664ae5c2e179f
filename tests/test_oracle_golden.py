"""Pins the CPU oracle against the reference's own known-answer tests:
the checksums carried by /root/reference/tests/inputs/*.perf (committed as data
in tests/golden/perf_golden.json by tools/make_golden.py).  Pass criterion is
the perf driver's own: |cs/ref - 1| <= threshold for both checksums
(tests/dbcsr_performance_multiply.F:658-676)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "perf_golden.json")))
CHECKED = sorted(k for k, v in CASES.items() if v["check"] == "T")


def test_all_nine_golden_cases_present():
    assert len(CHECKED) == 9


@pytest.mark.parametrize("name", CHECKED)
def test_perf_golden_checksums(name):
    c = CASES[name]
    assert c["data_type"] == 3 and c["symm_a"] == c["symm_b"] == c["symm_c"] == "N"
    assert c["limits"] == [0] * 6 and c["retain_sparsity"] == "F"
    A, B, Cm = O.perf_case(c["M"], c["N"], c["K"], c["sparsity_a"], c["sparsity_b"], c["sparsity_c"], c["bs_m"],
                           c["bs_n"], c["bs_k"], c["transa"], c["transb"])
    Cout, info = O.multiply(c["transa"], c["transb"], c["alpha"][0], A, B, c["beta"][0], Cm)
    cs, cs_pos = O.checksum(Cout), O.checksum(Cout, pos=True)
    assert abs(cs / c["checksum"] - 1.0) <= c["threshold"]
    assert abs(cs_pos / c["checksum_pos"] - 1.0) <= c["threshold"]
    # index structure sanity: sorted columns inside each row
    for r in range(Cout.nbr):
        cols = Cout.col_i[Cout.row_p[r]:Cout.row_p[r + 1]]
        assert np.all(np.diff(cols) > 0)


def test_block_counts_square_sparse():
    # 200x200 block grid, sparsity 0.9 -> C 3974, A 3942, B 4043 blocks
    A, B, Cm = O.perf_case(1000, 1000, 1000, 0.9, 0.9, 0.9, [1, 5], [1, 5], [1, 5])
    assert (Cm.nblks, A.nblks, B.nblks) == (3974, 3942, 4043)


def test_multiply_matches_dense():
    A, B, Cm = O.perf_case(230, 180, 150, 0.5, 0.6, 0.7, [1, 23, 2, 5], [1, 13], [2, 7, 1, 4])
    Cout, info = O.multiply("N", "N", 0.7, A, B, -1.3, Cm)
    ref = -1.3 * Cm.to_dense() + 0.7 * A.to_dense() @ B.to_dense()
    assert np.allclose(Cout.to_dense(), ref, rtol=1e-13, atol=1e-13)
    # flop count = sum over executed products of 2mnk (dbcsr_mm_csr.F:350)
    assert info["flop"] > 0


def test_multiply_transposes_and_retain_sparsity():
    A, B, Cm = O.perf_case(120, 90, 100, 0.4, 0.5, 0.6, [1, 5, 1, 3], [1, 4], [1, 6], transa="T", transb="T")
    Cout, _ = O.multiply("T", "T", 1.0, A, B, 1.0, Cm)
    ref = Cm.to_dense() + A.to_dense().T @ B.to_dense().T
    assert np.allclose(Cout.to_dense(), ref, rtol=1e-13, atol=1e-13)
    Cr, _ = O.multiply("T", "T", 1.0, A, B, 1.0, Cm, retain_sparsity=True)
    assert np.array_equal(Cr.row_p, Cm.row_p) and np.array_equal(Cr.col_i, Cm.col_i)
    mask = Cm.to_dense() != 0
    assert np.allclose(Cr.to_dense()[mask], ref[mask], rtol=1e-13)


def test_empty_and_ragged():
    # empty A -> C unchanged (beta=1), ragged tail block sizes
    A, B, Cm = O.perf_case(47, 31, 29, 1.0 - 1e-12, 0.3, 0.5, [1, 23], [1, 13], [1, 7])
    assert A.nblks == 0
    Cout, info = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
    assert info["flop"] == 0 and np.array_equal(Cout.col_i, Cm.col_i)
    assert np.array_equal(Cout.to_dense(), Cm.to_dense())
