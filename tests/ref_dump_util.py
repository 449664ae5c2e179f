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
    from oracle import oracle as O
    return O.perf_case(p["M"], p["N"], p["K"], p["sp"][0], p["sp"][1], p["sp"][2], p["bs_m"], p["bs_n"], p["bs_k"], p["transa"], p["transb"])


def oracle_run(p):
    from oracle import oracle as O
    A, B, Cm = oracle_inputs(p)
    eps = p["filter_eps"] if p["filter_eps"] >= 0 else 0.0
    if any(p["limits"]):
        return O.multiply_limits(p["transa"], p["transb"], p["alpha"], A, B, p["beta"], Cm, p["limits"], retain_sparsity=p["retain"], filter_eps=eps)
    return O.multiply(p["transa"], p["transb"], p["alpha"], A, B, p["beta"], Cm, retain_sparsity=p["retain"], filter_eps=eps)
