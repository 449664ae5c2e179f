#!/usr/bin/env python3
"""Generate tests/golden/*.json from the reference checkout (run in the build
container only; /root/reference does not exist on the GPU box).

 * perf_golden.json  -- parameters + golden checksums of the reference's own
   known-answer inputs tests/inputs/*.perf (format: tests/input.perf,
   parsed by tests/dbcsr_performance_multiply.F:106-165).
 * larnv_vectors.json -- LAPACK dlarnv/slarnv(idist=1) outputs for a few seeds,
   produced by the LAPACK bundled with scipy's OpenBLAS (scipy_dlarnv_).
Only data (inputs and expected outputs) is written; no reference source.
"""
import ctypes
import glob
import json
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def fnum(tok):
    return float(tok.lower().replace("d", "e"))


def parse_perf(path):
    toks = []
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        toks.append(line.split()[0])
    it = iter(toks)
    nx = lambda: next(it)
    d = {}
    d["npcols"] = int(nx())
    d["use_rma"] = nx()
    d["operation"] = nx()
    d["M"], d["N"], d["K"] = int(nx()), int(nx()), int(nx())
    d["sparsity_a"], d["sparsity_b"], d["sparsity_c"] = fnum(nx()), fnum(nx()), fnum(nx())
    d["transa"], d["transb"] = nx(), nx()
    d["symm_a"], d["symm_b"], d["symm_c"] = nx(), nx(), nx()
    d["data_type"] = int(nx())
    d["alpha"] = [fnum(nx()), fnum(nx())]
    d["beta"] = [fnum(nx()), fnum(nx())]
    d["limits"] = [int(nx()) for _ in range(6)]
    d["retain_sparsity"] = nx()
    d["nrep"] = int(nx())
    nm, nn, nk = int(nx()), int(nx()), int(nx())
    d["bs_m"] = [int(nx()) for _ in range(2 * nm)]
    d["bs_n"] = [int(nx()) for _ in range(2 * nn)]
    d["bs_k"] = [int(nx()) for _ in range(2 * nk)]
    d["check"] = nx()
    d["threshold"] = fnum(nx())
    d["checksum"] = fnum(nx())
    d["checksum_pos"] = fnum(nx())
    return d


def larnv_vectors():
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
    L = ctypes.CDLL(libs[0])
    out = []
    seeds = [[0, 0, 0, 1], [1, 2, 3, 5], [4095, 4095, 4095, 4095], [2029, 1153, 3541, 2047], [7, 42, 3, 43]]
    for s in seeds:
        for n in (1, 5, 64, 65, 130, 529):
            iseed = (ctypes.c_int * 4)(*s)
            x = np.empty(n, np.float64)
            L.scipy_dlarnv_(ctypes.byref(ctypes.c_int(1)), iseed, ctypes.byref(ctypes.c_int(n)),
                            x.ctypes.data_as(ctypes.c_void_p))
            iseed_s = (ctypes.c_int * 4)(*s)
            xs = np.empty(n, np.float32)
            L.scipy_slarnv_(ctypes.byref(ctypes.c_int(1)), iseed_s, ctypes.byref(ctypes.c_int(n)),
                            xs.ctypes.data_as(ctypes.c_void_p))
            out.append(dict(seed=s, n=n, d_hex=[float(v).hex() for v in x], d_seed_after=list(iseed),
                            s_hex=[float(v).hex() for v in xs], s_seed_after=list(iseed_s)))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = {}
    for p in sorted(glob.glob(os.path.join(REF, "tests", "inputs", "*.perf"))):
        cases[os.path.basename(p)] = parse_perf(p)
    json.dump(cases, open(os.path.join(OUT, "perf_golden.json"), "w"), indent=1)
    json.dump(larnv_vectors(), open(os.path.join(OUT, "larnv_vectors.json"), "w"))
    print("wrote", len(cases), "perf cases")


if __name__ == "__main__":
    sys.exit(main())
