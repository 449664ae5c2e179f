#!/usr/bin/env python3
"""Generates tests/golden/ref_dump.json: outputs of the REAL reference library (true dbcsr_multiply on the CPU/BLAS
path, built unchanged by tools/build_dbcsr_host.py) for multiply cases that no .perf golden file covers --
beta = 0, retain_sparsity, filter_eps, submatrix limits, transposes, mixed block sizes, symmetric matrices.

Build-container tool (the reference is not on the GPU box).  Only data is written: the case parameters, and for
each case the block index of C, flop, checksums and (small cases) all values, as produced by
oracle/_ref/host_cpu/dbcsr_ref_dump (tests/fortran/dbcsr_ref_dump.F90, this repository's own driver program).
"""
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "oracle", "_ref", "host_cpu", "dbcsr_ref_dump")

D = dict(transa="N", transb="N", symm_a="N", symm_b="N", symm_c="N", alpha=1.0, beta=1.0, limits=[0] * 6, retain=False,
         filter_eps=-1.0, values=True, data_type=3)


def case(name, M, N, K, sp, bs_m, bs_n=None, bs_k=None, **kw):
    c = dict(D)
    c.update(name=name, M=M, N=N, K=K, sp=list(sp), bs_m=bs_m, bs_n=bs_n or bs_m, bs_k=bs_k or bs_m)
    c.update(kw)
    return c


CASES = [
    case("basic_5", 100, 100, 100, (0.5, 0.5, 0.5), [1, 5]),
    case("beta0_wipes_c", 100, 100, 100, (0.7, 0.7, 0.3), [1, 5], beta=0.0),
    case("beta0_retain", 100, 100, 100, (0.7, 0.7, 0.3), [1, 5], beta=0.0, retain=True),
    case("beta0_alpha0", 60, 60, 60, (0.5, 0.5, 0.5), [1, 4], alpha=0.0, beta=0.0),
    case("alpha0_beta1", 60, 60, 60, (0.5, 0.5, 0.5), [1, 4], alpha=0.0, beta=1.0),
    case("alpha_beta_mixed_TN", 90, 84, 96, (0.5, 0.6, 0.7), [1, 13, 2, 5], [1, 23, 1, 4], [2, 7, 1, 32], alpha=-0.5, beta=2.0, transa="T"),
    case("mixed_NT", 90, 84, 96, (0.5, 0.6, 0.7), [1, 13, 2, 5], [1, 23, 1, 4], [2, 7, 1, 32], alpha=0.7, beta=1.3, transb="T"),
    case("mixed_TT", 90, 84, 96, (0.5, 0.6, 0.7), [1, 13, 2, 5], [1, 23, 1, 4], [2, 7, 1, 32], transa="T", transb="T"),
    case("retain", 120, 120, 120, (0.6, 0.6, 0.8), [1, 13, 1, 23], [1, 23, 1, 32], [1, 5, 1, 13], retain=True),
    case("filter_eps_small", 100, 100, 100, (0.7, 0.7, 0.9), [1, 5], filter_eps=1.0),
    case("filter_eps_mid", 100, 100, 100, (0.7, 0.7, 0.9), [1, 5], filter_eps=12.0),
    case("filter_eps_large", 100, 100, 100, (0.7, 0.7, 0.9), [1, 5], filter_eps=40.0),
    case("filter_eps_mixed", 110, 90, 100, (0.6, 0.6, 0.8), [1, 3, 1, 7], [1, 5, 1, 2], [1, 4, 1, 6], filter_eps=6.0, alpha=0.5, beta=1.0),
    case("filter_eps_retain", 100, 100, 100, (0.7, 0.7, 0.5), [1, 5], filter_eps=12.0, retain=True),
    case("limits_cut_new", 40, 36, 44, (0.6, 0.6, 0.7), [1, 5, 1, 3], [1, 4], [1, 7, 1, 2], alpha=0.5, beta=2.0, limits=[7, 29, 6, 31, 10, 33]),
    case("limits_beta0", 50, 50, 50, (0.5, 0.5, 0.5), [1, 2], beta=0.0, limits=[9, 18, 11, 20, 1, 50]),
    case("limits_beta0_k", 50, 50, 50, (0.5, 0.5, 0.5), [1, 2], beta=0.0, limits=[1, 50, 1, 50, 9, 18]),
    case("limits_retain", 25, 50, 75, (0.5, 0.5, 0.5), [1, 2], [1, 2, 1, 3], [1, 3, 1, 2], beta=0.0, retain=True, limits=[11, 20, 11, 20, 6, 10]),
    case("limits_T", 40, 36, 44, (0.6, 0.6, 0.7), [1, 5, 1, 3], [1, 4], [1, 7, 1, 2], transa="T", transb="T", limits=[7, 29, 6, 31, 10, 33]),
    case("config3_like", 13 + 23 + 32 + 13 + 23 + 24, 128, 128, (0.4, 0.4, 0.6), [1, 13, 1, 23, 1, 32]),
    case("h2o_like_23", 23 * 8 + 16, 200, 200, (0.7, 0.7, 0.7), [1, 23], values=False),
    case("singleblock", 50, 20, 10, (0.0, 0.0, 0.0), [1, 50], [1, 20], [1, 10]),
    case("empty_a", 47, 31, 29, (1.0 - 1e-12, 0.3, 0.5), [1, 23], [1, 13], [1, 7], beta=2.0),
    # symmetric matrices (dbcsr_test_multiply.F:95-115 runs the symmetry combinations; square, equal block sizes)
    case("symm_a_S", 60, 60, 60, (0.5, 0.5, 0.5), [1, 4], symm_a="S"),
    case("symm_b_S", 60, 60, 60, (0.5, 0.5, 0.5), [1, 4], symm_b="S"),
    case("symm_ab_S", 60, 60, 60, (0.5, 0.5, 0.5), [1, 3, 1, 5], symm_a="S", symm_b="S", alpha=0.5, beta=2.0),
    case("symm_a_A", 60, 60, 60, (0.5, 0.5, 0.5), [1, 4], symm_a="A"),
    case("symm_a_S_T", 60, 60, 60, (0.5, 0.5, 0.5), [1, 4], symm_a="S", transa="T"),
    case("symm_b_A_T", 60, 60, 60, (0.5, 0.5, 0.5), [1, 4], symm_b="A", transb="T"),
    # symmetric / antisymmetric PRODUCT matrix: only the blocks of its stored (canonical) form are computed
    # (dbcsr_test_multiply.F:187-200 runs these with (N, T) / (T, N) and full limits)
    case("symm_c_S_NT", 60, 60, 50, (0.5, 0.5, 0.6), [1, 4], [1, 4], [1, 5], symm_c="S", transb="T", alpha=0.5, beta=2.0),
    case("symm_c_S_TN", 60, 60, 50, (0.5, 0.5, 0.6), [1, 4, 1, 6], [1, 4, 1, 6], [1, 5], symm_c="S", transa="T"),
    case("symm_c_S_beta0", 60, 60, 50, (0.5, 0.5, 0.3), [1, 4], [1, 4], [1, 5], symm_c="S", transb="T", beta=0.0),
    case("symm_c_S_retain", 60, 60, 50, (0.5, 0.5, 0.5), [1, 4], [1, 4], [1, 5], symm_c="S", transb="T", retain=True),
    case("symm_c_S_filter", 100, 100, 100, (0.7, 0.7, 0.9), [1, 5], symm_c="S", transb="T", filter_eps=12.0),
    case("symm_c_A_NT", 60, 60, 50, (0.5, 0.5, 0.6), [1, 4], [1, 4], [1, 5], symm_c="A", transb="T", alpha=-1.5, beta=0.5),
    case("symm_abc_S", 60, 60, 60, (0.5, 0.5, 0.5), [1, 3, 1, 5], symm_a="S", symm_b="S", symm_c="S"),
    case("symm_c_S_NN", 48, 48, 48, (0.5, 0.5, 0.5), [1, 4], symm_c="S"),
    # single precision (data_type 1 of the .perf format; BASELINE config 5's shape: 32 x 32 blocks, and a mixed one)
    case("fp32_32", 32 * 6, 32 * 5, 32 * 7, (0.6, 0.6, 0.7), [1, 32], data_type=1),
    case("fp32_32_beta0_T", 32 * 5, 32 * 5, 32 * 6, (0.5, 0.5, 0.5), [1, 32], data_type=1, beta=0.0, transa="T"),
    case("fp32_mixed", 120, 110, 130, (0.5, 0.6, 0.7), [1, 13, 1, 32, 1, 7], [1, 23, 1, 32], [1, 13, 1, 32, 1, 9], data_type=1, alpha=0.5, beta=2.0),
    case("fp32_filter_retain", 100, 100, 100, (0.7, 0.7, 0.5), [1, 5], data_type=1, filter_eps=12.0, retain=True),
]


def fstr(x):
    return ("%.17g" % x).replace("e", "d") if "e" in "%.17g" % x else "%.17gd0" % x


def write_nml(c, path):
    with open(path, "w") as f:
        f.write("&spec\n m=%d, n=%d, k=%d,\n sp_a=%s, sp_b=%s, sp_c=%s,\n" % (c["M"], c["N"], c["K"], *map(fstr, c["sp"])))
        f.write(" transa='%s', transb='%s', symm_a='%s', symm_b='%s', symm_c='%s',\n" % (c["transa"], c["transb"], c["symm_a"], c["symm_b"], c["symm_c"]))
        f.write(" alpha=%s, beta=%s, retain=%s, filter_eps=%s,\n" % (fstr(c["alpha"]), fstr(c["beta"]), ".TRUE." if c["retain"] else ".FALSE.", fstr(c["filter_eps"])))
        f.write(" limits=%s,\n" % ",".join(map(str, c["limits"])))
        for d in ("m", "n", "k"):
            bs = c["bs_" + d]
            f.write(" nbs_%s=%d, bs_%s=%s,\n" % (d, len(bs) // 2, d, ",".join(map(str, bs))))
        f.write(" dump_values=%d, data_type=%d\n/\n" % (1 if c["values"] else 0, c.get("data_type", 3)))


def parse_dump(path):
    out = {"blocks": []}
    toks = open(path).read().split("\n")
    i = 0
    while i < len(toks):
        t = toks[i].split()
        i += 1
        if not t:
            continue
        if t[0] == "dims":
            out["nblkrows"], out["nblkcols"], out["nblks"] = map(int, t[1:4])
        elif t[0] == "flop":
            out["flop"] = int(t[1])
        elif t[0] in ("checksum", "checksum_a", "checksum_b"):
            out[t[0]] = [float(t[1]), float(t[2])]
        elif t[0] == "block":
            b = {"row": int(t[1]), "col": int(t[2]), "tr": t[3] == "T", "m": int(t[4]), "n": int(t[5])}
            vals = []
            while i < len(toks) and toks[i].strip() and not toks[i].lstrip()[0].isalpha():
                vals += [float(x) for x in toks[i].split()]
                i += 1
            if vals:
                assert len(vals) == b["m"] * b["n"], (b, len(vals))
                b["v"] = vals
            out["blocks"].append(b)
    blocks = sorted(out.pop("blocks"), key=lambda b: (b["row"], b["col"]))
    assert len(blocks) == out["nblks"]
    # compact form: parallel arrays in (row, col) order; values of all blocks concatenated (column-major as stored), base64 of float64
    for key in ("row", "col", "m", "n"):
        out[key] = [b[key] for b in blocks]
    out["tr"] = [int(b["tr"]) for b in blocks]
    if blocks and "v" in blocks[0]:
        import base64
        import struct
        flat = [v for b in blocks for v in b["v"]]
        out["values_b64"] = base64.b64encode(struct.pack("<%dd" % len(flat), *flat)).decode()
    return out


def run_case(c, exe=DUMP, env=None, with_stdout=False):
    with tempfile.TemporaryDirectory() as td:
        nml, out = os.path.join(td, "case.nml"), os.path.join(td, "out.txt")
        write_nml(c, nml)
        e = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="2")
        e.update(env or {})
        r = subprocess.run([exe, nml, out], cwd=td, env=e, capture_output=True, text=True, timeout=600)
        if r.returncode != 0 or not os.path.exists(out):
            raise RuntimeError("dbcsr_ref_dump failed on %s (return code %d):\n%s\n%s" % (c["name"], r.returncode, r.stdout[-2000:], r.stderr[-2000:]))
        return (parse_dump(out), r.stdout) if with_stdout else parse_dump(out)


def main():
    if not os.path.exists(DUMP):
        raise SystemExit("build the reference host first: python tools/build_dbcsr_host.py cpu")
    res = {}
    for c in CASES:
        d = run_case(c)
        res[c["name"]] = {"params": c, "result": d}
        print("%-22s C blocks %5d  flop %10d  checksum %.15e" % (c["name"], d["nblks"], d["flop"], d["checksum"][0]))
    path = os.path.join(ROOT, "tests", "golden", "ref_dump.json")
    with open(path, "w") as f:
        json.dump(res, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
