"""Loader for tests/golden/ref_dump.json (outputs of the real reference library, tools/make_ref_fixtures.py)."""
import base64
import json
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_dump.json")
_CACHE = {}


def cases():
    if "d" not in _CACHE:
        _CACHE["d"] = json.load(open(PATH))
    return _CACHE["d"]


def names(pred=lambda p: True):
    return sorted(k for k, v in cases().items() if pred(v["params"]))


def np_dtype(p):
    """data_type of the case as in the .perf format: 1 = real(4), 3 = real(8)"""
    return np.float32 if p.get("data_type", 3) == 1 else np.float64


def nonsymmetric(p):
    return p["symm_a"] == "N" and p["symm_b"] == "N" and p["symm_c"] == "N"


class RefResult:
    """C of the reference in index order: row_p / col_i (0-based), per-block (m, n), values concatenated column-major."""

    def __init__(self, name):
        c = cases()[name]
        self.params, r = c["params"], c["result"]
        self.flop, self.checksum, self.checksum_pos = r["flop"], r["checksum"][0], r["checksum"][1]
        self.nblkrows, self.nblks = r["nblkrows"], r["nblks"]
        self.rows = np.asarray(r["row"], np.int64) - 1
        self.col_i = np.asarray(r["col"], np.int32) - 1
        self.m, self.n, self.tr = np.asarray(r["m"], np.int64), np.asarray(r["n"], np.int64), np.asarray(r["tr"], bool)
        self.row_p = np.zeros(self.nblkrows + 1, np.int32)
        np.add.at(self.row_p, self.rows + 1, 1)
        self.row_p = np.cumsum(self.row_p).astype(np.int32)
        self.data = np.frombuffer(base64.b64decode(r["values_b64"]), "<f8") if "values_b64" in r else None
        self.checksum_a, self.checksum_b = r["checksum_a"], r["checksum_b"]


def oracle_inputs(p):
    """(A, B, C) as the reference's generator makes them for the case (C, A, B in this order: the matrix counter advances);
    a symmetric / antisymmetric operand is returned as the FULL matrix (desymmetrized), its stored triangle as 4th / 5th item."""
    from oracle import oracle as O
    if nonsymmetric(p):
        return O.perf_case(p["M"], p["N"], p["K"], p["sp"][0], p["sp"][1], p["sp"][2], p["bs_m"], p["bs_n"], p["bs_k"], p["transa"], p["transb"],
                           dtype=np_dtype(p))
    assert np_dtype(p) == np.float64
    sm, sn, sk = O.make_block_sizes(p["M"], p["bs_m"]), O.make_block_sizes(p["N"], p["bs_n"]), O.make_block_sizes(p["K"], p["bs_k"])
    c0 = O.RANDMAT_SEED_INIT
    Cm = O.make_random_matrix(sm, sn, p["sp"][2], c0 + 1) if p["symm_c"] == "N" else O.make_random_matrix_symmetric(sm, p["sp"][2], c0 + 1, p["symm_c"])

    def operand(symm, rs, cs, sp, counter):
        if symm == "N":
            return O.make_random_matrix(rs, cs, sp, counter)
        assert np.array_equal(rs, cs)
        return O.desymmetrize(O.make_random_matrix_symmetric(rs, sp, counter, symm), symm)

    A = operand(p["symm_a"], *((sk, sm) if p["transa"] != "N" else (sm, sk)), p["sp"][0], c0 + 2)
    B = operand(p["symm_b"], *((sn, sk) if p["transb"] != "N" else (sk, sn)), p["sp"][1], c0 + 3)
    return A, B, Cm


def stored_operands(p):
    """the stored (one-triangle) form of the symmetric operands of a case: {"A": (Bcsr, symmetry), ...}"""
    from oracle import oracle as O
    sm, sn, sk = O.make_block_sizes(p["M"], p["bs_m"]), O.make_block_sizes(p["N"], p["bs_n"]), O.make_block_sizes(p["K"], p["bs_k"])
    c0 = O.RANDMAT_SEED_INIT
    out = {}
    if p["symm_a"] != "N":
        out["A"] = (O.make_random_matrix_symmetric(sm, p["sp"][0], c0 + 2, p["symm_a"]), p["symm_a"])
    if p["symm_b"] != "N":
        out["B"] = (O.make_random_matrix_symmetric(sk, p["sp"][1], c0 + 3, p["symm_b"]), p["symm_b"])
    return out


def oracle_run(p):
    from oracle import oracle as O
    A, B, Cm = oracle_inputs(p)
    if np_dtype(p) == np.float32:   # the oracle's multiply is double precision: single-precision inputs are widened (exactly)
        wide = lambda M: O.Bcsr(M.row_sizes, M.col_sizes, M.row_p, M.col_i, M.blk_p, M.data.astype(np.float64))
        A, B, Cm = wide(A), wide(B), wide(Cm)
    eps = p["filter_eps"] if p["filter_eps"] >= 0 else 0.0
    if p["symm_c"] != "N":
        assert not any(p["limits"])
        return O.multiply(p["transa"], p["transb"], p["alpha"], A, B, p["beta"], Cm, retain_sparsity=p["retain"], filter_eps=eps, c_symmetry=p["symm_c"])
    if any(p["limits"]):
        return O.multiply_limits(p["transa"], p["transb"], p["alpha"], A, B, p["beta"], Cm, p["limits"], retain_sparsity=p["retain"], filter_eps=eps)
    return O.multiply(p["transa"], p["transb"], p["alpha"], A, B, p["beta"], Cm, retain_sparsity=p["retain"], filter_eps=eps)
