"""Row f4 of SURVEY.md section 8: the UNCHANGED reference Fortran host (DBCSR library + its own drivers, expanded with
tools/fypp_lite.py and compiled with amdflang -D__DBCSR_ACC by tools/build_dbcsr_host.py) linked against this
repository's libdbcsr_acc_amd.so, run on the GPU through the true dbcsr_multiply:
  * the reference's performance driver on its own golden .perf inputs (it checks its checksums itself),
  * the reference's unit tests dbcsr_unittest1 / dbcsr_unittest3 (GPU block-size mixes),
  * this repository's dump driver on the cases of tests/golden/ref_dump.json: what the acc back end produces under the
    real host equals what the reference's BLAS path produced (index identical, values 1e-10).
The binaries are built in the build container (oracle/_ref/host_acc, git-ignored, they travel with the snapshot)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests import ref_dump_util as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "oracle", "_ref", "host_acc")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "perf_golden.json")))
ENV = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="4")

needs_host = pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "dbcsr_perf")),
                                reason="reference Fortran host not built (tools/build_dbcsr_host.py acc)")



# "... PASSED !" lines of the reference's unit test programs (counted on the unchanged CPU build, oracle/_ref/host_cpu)
EXPECTED_PASSED = {"dbcsr_unittest1": 8124, "dbcsr_unittest3": 756}


def check_unittest_output(out, prog):
    """the reference's multiply tests print one "... PASSED !" line per case and "... FAILED !" before aborting
    (tests/dbcsr_test_multiply.F:497-511): every case must have passed, none failed, and nothing may report an error"""
    up = out.upper()
    assert " FAILED !" not in up, out[-3000:]
    assert up.count("PASSED !") == EXPECTED_PASSED[prog], (up.count("PASSED !"), EXPECTED_PASSED[prog])
    assert "ERROR" not in up.replace("ERROR_TOLERANCE", ""), out[-3000:]

def write_perf(c, path):
    def d(x):  # the reference's ator() wants a decimal point: 1.0d0, not 1d0
        m, _, e = ("%.17e" % x).partition("e")
        m = m.rstrip("0")
        return "%s%sd%d" % (m, "0" if m.endswith(".") else "", int(e))
    toks = [c["npcols"], c["use_rma"], c["operation"], c["M"], c["N"], c["K"], d(c["sparsity_a"]), d(c["sparsity_b"]), d(c["sparsity_c"]),
            c["transa"], c["transb"], c["symm_a"], c["symm_b"], c["symm_c"], c["data_type"], d(c["alpha"][0]), d(c["alpha"][1]),
            d(c["beta"][0]), d(c["beta"][1]), *c["limits"], c["retain_sparsity"], min(int(c["nrep"]), 2),
            len(c["bs_m"]) // 2, len(c["bs_n"]) // 2, len(c["bs_k"]) // 2, *c["bs_m"], *c["bs_n"], *c["bs_k"],
            c["check"], "%.3E" % c["threshold"], "%.15E" % c["checksum"], "%.15E" % c["checksum_pos"]]
    with open(path, "w") as f:
        f.write("\n".join(str(t) for t in toks) + "\n")


@needs_host
@pytest.mark.parametrize("name", sorted(k for k, v in GOLD.items() if v["check"] == "T" and v["data_type"] == 3))
def test_reference_perf_driver_through_acc_backend(name, tmp_path):
    c = GOLD[name]
    write_perf(c, tmp_path / "case.perf")
    r = subprocess.run([os.path.join(HOST, "dbcsr_perf"), str(tmp_path / "case.perf")], cwd=tmp_path, env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    m = re.search(r"checksum\(C_out\)\s*=\s*([0-9.E+-]+)", r.stdout)
    mp = re.search(r"checksum\(C_out\) POS\s*=\s*([0-9.E+-]+)", r.stdout)
    assert m and mp, r.stdout[-3000:]
    assert abs(float(m.group(1)) / c["checksum"] - 1.0) <= c["threshold"]
    assert abs(float(mp.group(1)) / c["checksum_pos"] - 1.0) <= c["threshold"]
    # the stacks really ran on the accelerator back end (DBCSR's own statistics: share of flops by driver)
    acc = re.search(r"flops total\s+\S+\s+([0-9.]+)%\s+([0-9.]+)%\s+([0-9.]+)%", r.stdout)
    assert acc, r.stdout[-3000:]
    if max(c["bs_m"][1::2] + c["bs_n"][1::2] + c["bs_k"][1::2]) <= 80:
        assert float(acc.group(3)) > 50.0, "less than half of the flops went through libsmm_acc_process:\n" + r.stdout[-3000:]


@needs_host
@pytest.mark.parametrize("prog", ["dbcsr_unittest1", "dbcsr_unittest3"])
def test_reference_unittests_through_acc_backend(prog, tmp_path):
    r = subprocess.run([os.path.join(HOST, prog)], cwd=tmp_path, env=ENV, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    check_unittest_output(r.stdout, prog)


@needs_host
@pytest.mark.parametrize("name", R.names(lambda p: p["values"]))
def test_dump_driver_through_acc_backend(name):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ref_fixtures as F
    ref = R.RefResult(name)
    got = F.run_case(ref.params, exe=os.path.join(HOST, "dbcsr_ref_dump"), env={"OMP_NUM_THREADS": "4"})
    assert got["nblks"] == ref.nblks and got["flop"] == ref.flop
    assert np.array_equal(np.asarray(got["row"]) - 1, ref.rows) and np.array_equal(np.asarray(got["col"], np.int32) - 1, ref.col_i)
    import base64
    data = np.frombuffer(base64.b64decode(got["values_b64"]), "<f8") if ref.nblks else np.zeros(0)
    scale = max(np.max(np.abs(ref.data)), 1e-300) if ref.nblks else 1.0
    assert data.size == ref.data.size and np.max(np.abs(data - ref.data), initial=0.0) <= (5e-6 if R.np_dtype(ref.params) == np.float32 else 1e-10) * scale


# ---- the unchanged host's G2G variant (DBCSR_USE_ACC_G2G=1: src/mm/dbcsr_mm.F:909-922 -> multiply_cannon_g2g,
# dbcsr_mm_cannon.F:1773-2836): panels uploaded once and kept as device pointers, block norms computed ON THE DEVICE by
# c_calculate_norms (its only in-situ caller, src/mm/dbcsr_mm_common.F:117-130), four OpenMP threads driving the acc ABI at once --
ENV_G2G = dict(ENV, DBCSR_USE_ACC_G2G="1")   # (logical parameters are read as integers, src/core/dbcsr_config.F:306-318)


@needs_host
@pytest.mark.parametrize("name", sorted(k for k, v in GOLD.items() if v["check"] == "T" and v["data_type"] == 3))
def test_reference_perf_driver_through_acc_backend_g2g(name, tmp_path):
    c = GOLD[name]
    write_perf(c, tmp_path / "case.perf")
    r = subprocess.run([os.path.join(HOST, "dbcsr_perf"), str(tmp_path / "case.perf")], cwd=tmp_path, env=ENV_G2G, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert re.search(r"ACC: Use G2G algorithm\s+T", r.stdout), "the host did not switch to the G2G algorithm:\n" + r.stdout[:3000]
    m = re.search(r"checksum\(C_out\)\s*=\s*([0-9.E+-]+)", r.stdout)
    mp = re.search(r"checksum\(C_out\) POS\s*=\s*([0-9.E+-]+)", r.stdout)
    assert m and mp, r.stdout[-3000:]
    assert abs(float(m.group(1)) / c["checksum"] - 1.0) <= c["threshold"]
    assert abs(float(mp.group(1)) / c["checksum_pos"] - 1.0) <= c["threshold"]


@needs_host
@pytest.mark.parametrize("name", R.names(lambda p: p["values"] and R.np_dtype(p) == np.float64))
def test_dump_driver_through_acc_backend_g2g(name):
    """the double-precision reference dumps (filter_eps cases included: their block norms now come from c_calculate_norms on
    the device) through the unchanged host's G2G algorithm"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import base64
    import make_ref_fixtures as F
    ref = R.RefResult(name)
    got = F.run_case(ref.params, exe=os.path.join(HOST, "dbcsr_ref_dump"), env={"OMP_NUM_THREADS": "4", "DBCSR_USE_ACC_G2G": "1"})
    assert got["nblks"] == ref.nblks and got["flop"] == ref.flop
    assert np.array_equal(np.asarray(got["row"]) - 1, ref.rows) and np.array_equal(np.asarray(got["col"], np.int32) - 1, ref.col_i)
    data = np.frombuffer(base64.b64decode(got["values_b64"]), "<f8") if ref.nblks else np.zeros(0)
    scale = max(np.max(np.abs(ref.data)), 1e-300) if ref.nblks else 1.0
    assert data.size == ref.data.size and np.max(np.abs(data - ref.data), initial=0.0) <= 1e-10 * scale


# ---- the call-site change of INTEGRATION.md section 2, compiled into the reference (tools/build_dbcsr_host.py resident):
# dbcsr_multiply of the otherwise unchanged library hands the whole multiply to the device-resident engine -------------------
HOST_RES = os.path.join(ROOT, "oracle", "_ref", "host_resident")
needs_resident = pytest.mark.skipif(not os.path.exists(os.path.join(HOST_RES, "dbcsr_perf")),
                                    reason="patched reference host not built (tools/build_dbcsr_host.py resident)")
ENV_RES = dict(ENV, DBCSR_AMD_RESIDENT="1")


@needs_resident
@pytest.mark.parametrize("name", sorted(k for k, v in GOLD.items() if v["check"] == "T" and v["data_type"] == 3))
def test_reference_perf_driver_through_resident_engine(name, tmp_path):
    c = GOLD[name]
    write_perf(c, tmp_path / "case.perf")
    r = subprocess.run([os.path.join(HOST_RES, "dbcsr_perf"), str(tmp_path / "case.perf")], cwd=tmp_path, env=ENV_RES, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]   # the driver checks its golden checksums itself (1e-11)
    m = re.search(r"checksum\(C_out\)\s*=\s*([0-9.E+-]+)", r.stdout)
    assert m and abs(float(m.group(1)) / c["checksum"] - 1.0) <= c["threshold"]
    # no parameter stack was ever built: the reference's own multiplication statistics stay empty
    mm = re.search(r"matmuls total\s+(\d+)", r.stdout)
    assert mm and int(mm.group(1)) == 0, r.stdout[-3000:]


@needs_resident
@pytest.mark.parametrize("name", R.names(lambda p: p["values"]))
def test_dump_driver_through_resident_engine(name):
    """every case of the reference dumps -- single precision, symmetric / antisymmetric operands and product matrices included --
    through the patched host: dbcsr_multiply hands the multiply to the device-resident engine"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import base64
    import make_ref_fixtures as F
    ref = R.RefResult(name)
    got, stdout = F.run_case(ref.params, exe=os.path.join(HOST_RES, "dbcsr_ref_dump"), env={"OMP_NUM_THREADS": "4", "DBCSR_AMD_RESIDENT": "1v"},
                             with_stdout=True)
    assert "dbcsr_amd_resident:" in stdout, "the multiply did not take the device-resident path:\n" + stdout[-1500:]
    assert got["nblks"] == ref.nblks and got["flop"] == ref.flop
    assert np.array_equal(np.asarray(got["row"]) - 1, ref.rows) and np.array_equal(np.asarray(got["col"], np.int32) - 1, ref.col_i)
    data = np.frombuffer(base64.b64decode(got["values_b64"]), "<f8") if ref.nblks else np.zeros(0)
    scale = max(np.max(np.abs(ref.data)), 1e-300) if ref.nblks else 1.0
    assert data.size == ref.data.size and np.max(np.abs(data - ref.data), initial=0.0) <= (5e-6 if R.np_dtype(ref.params) == np.float32 else 1e-10) * scale


@needs_resident
@pytest.mark.parametrize("prog", ["dbcsr_unittest1", "dbcsr_unittest3"])
def test_reference_unittests_through_resident_engine(prog, tmp_path):
    """the reference's own multiply unit tests (all symmetry / transposition / limit / type combinations of tests/dbcsr_test_multiply.F) on the
    patched host: what the glue accepts (real(8), with or without symmetry) runs on the device-resident engine, the rest falls through
    to the reference path; the program checks every result against its dense computation itself."""
    r = subprocess.run([os.path.join(HOST_RES, prog)], cwd=tmp_path, env=dict(ENV_RES, DBCSR_AMD_RESIDENT="1v"), capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    check_unittest_output(r.stdout, prog)
    assert r.stdout.count("dbcsr_amd_resident:") > 10, "hardly any multiply took the device-resident path"


@pytest.mark.skipif(not os.path.exists(os.path.join(HOST_RES, "dbcsr_resident_loop")), reason="patched reference host not built (tools/build_dbcsr_host.py resident)")
@pytest.mark.parametrize("args", [("2316", "0.8", "23", "5"), ("3000", "0.7", "13", "3"), ("1500", "0.5", "32", "4")], ids=["23", "13", "32"])
def test_fortran_host_keeps_matrices_on_the_device(args, tmp_path):
    """tests/fortran/dbcsr_resident_loop.F90: a Fortran host that uploads A, B, C once (dbcsr_amd_dev_create), multiplies nrep times in
    HBM (dbcsr_amd_dev_multiply: C <- beta C + alpha A B, C's pattern changes after the first step, the plan is reused afterwards) and
    downloads C once, against the same loop through the library's own dbcsr_multiply: same checksums (1e-10), and no PCIe traffic per
    multiply -- the per-multiply time of the resident loop must be far below the build's own dbcsr_multiply."""
    r = subprocess.run([os.path.join(HOST_RES, "dbcsr_resident_loop"), *args, "1"], cwd=tmp_path, env=dict(ENV, DBCSR_AMD_RESIDENT="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    d = re.search(r"relative difference\s+([0-9.E+-]+)", r.stdout)
    assert d and float(d.group(1)) <= 1e-10, r.stdout[-2000:]
    t_dev = float(re.search(r"per multiply \[s\]\s+([0-9.]+)", r.stdout).group(1))
    t_ref = float(re.search(r"dbcsr_multiply of this build, per multiply \[s\]\s+([0-9.]+)", r.stdout).group(1))
    assert t_dev < t_ref, r.stdout[-2000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(HOST_RES, "dbcsr_resident_loop")), reason="patched reference host not built (tools/build_dbcsr_host.py resident)")
@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("args", [("2316", "0.8", "23", "5"), ("1500", "0.5", "32", "4")], ids=["23", "32"])
def test_fortran_host_out_of_place_products_and_product_as_operand(args, mode, tmp_path):
    """round 5: mode 1 -- every multiply starts from the ORIGINAL C and writes its product to a second resident
    matrix (dbcsr_amd_dev_multiply, c_out): A, B and C_in are the same generation of the same device arrays in every call, so the engine
    reuses its plan by address (dbcsr_amd_resident.F stamps every index array it owns); mode 2 -- the product of one multiply is the
    LEFT operand of the next (dbcsr_amd_dev_as_operand), nothing goes through a dbcsr_type in between.  Checksums of the same loops
    through the library's own dbcsr_multiply, 1e-10."""
    r = subprocess.run([os.path.join(HOST_RES, "dbcsr_resident_loop"), *args, "1", mode], cwd=tmp_path, env=dict(ENV, DBCSR_AMD_RESIDENT="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "resident_loop: mode %s" % mode in r.stdout, r.stdout[-2000:]
    d = re.search(r"relative difference\s+([0-9.E+-]+)", r.stdout)
    assert d and float(d.group(1)) <= 1e-10, r.stdout[-2000:]
