"""The UNCHANGED reference as a real multi-rank MPI program (tools/build_dbcsr_host.py cpu_mpi / acc_mpi: -D__parallel, the image's
MPICH through a module `mpi` made of MPICH's own mpif.h, launched with mpiexec):
  * CPU build, 2 / 4 ranks, the reference's golden .perf inputs (the driver checks its checksums itself): pins what a multi-rank
    run of the reference computes -- and that the MPI build of this repository's tooling is a faithful one;
  * (-m gpu) the same library linked against libdbcsr_acc_amd.so, 2 / 4 / 8 ranks SHARING the one GPU of the box (device = rank
    mod ndevices, src/core/dbcsr_lib.F:231-236): the reference's own Fortran Cannon loop -- make_m2s, MPI isend / irecv of the
    panels, one OpenMP team per rank driving the acc ABI -- on this back end, against the same golden checksums.
The binaries are built in the build container (oracle/_ref/host_cpu_mpi, host_acc_mpi: git-ignored, they travel with the snapshot)."""
import json
import os
import re
import shutil
import subprocess

import pytest

from tests.test_gpu_fortran_host import write_perf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "perf_golden.json")))
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
ENV = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="2")
CASES = sorted(k for k, v in GOLD.items() if v["check"] == "T" and v["data_type"] == 3)


def run_perf(host, name, nranks, tmp_path, env):
    exe = os.path.join(ROOT, "oracle", "_ref", host, "dbcsr_perf")
    if not (os.path.exists(exe) and os.path.exists(MPIEXEC)):
        pytest.skip("%s or mpiexec not available" % host)
    c = dict(GOLD[name])
    c["npcols"] = 0   # let MPI_Dims_create choose the grid (the golden checksums do not depend on it)
    write_perf(c, tmp_path / "case.perf")
    r = subprocess.run([MPIEXEC, "-n", str(nranks), exe, str(tmp_path / "case.perf")], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert re.search(r"numnodes\s+%d\b" % nranks, r.stdout), r.stdout[:2000]
    m = re.search(r"checksum\(C_out\)\s*=\s*([0-9.E+-]+)", r.stdout)
    mp = re.search(r"checksum\(C_out\) POS\s*=\s*([0-9.E+-]+)", r.stdout)
    assert m and mp, r.stdout[-3000:]
    assert abs(float(m.group(1)) / c["checksum"] - 1.0) <= c["threshold"]
    assert abs(float(mp.group(1)) / c["checksum_pos"] - 1.0) <= c["threshold"]
    return r.stdout


@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("name", ["test_square_sparse.perf", "test_rect1_sparse.perf", "test_rect2_dense.perf"])
def test_reference_mpi_build_reproduces_golden_checksums_on_cpu(name, nranks, tmp_path):
    run_perf("host_cpu_mpi", name, nranks, tmp_path, ENV)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4, 8])
@pytest.mark.parametrize("name", CASES)
def test_reference_mpi_host_on_acc_backend_shared_gpu(name, nranks, tmp_path):
    if nranks == 8 and name not in ("test_square_sparse.perf", "test_square_dense.perf", "test_rect1_sparse.perf"):
        pytest.skip("eight ranks on the three cases with enough blocks per rank")
    out = run_perf("host_acc_mpi", name, nranks, tmp_path, ENV)
    c = GOLD[name]
    acc = re.search(r"flops total\s+\S+\s+([0-9.]+)%\s+([0-9.]+)%\s+([0-9.]+)%", out)
    assert acc, out[-3000:]
    if max(c["bs_m"][1::2] + c["bs_n"][1::2] + c["bs_k"][1::2]) <= 80:
        assert float(acc.group(3)) > 50.0, "less than half of the flops went through libsmm_acc_process:\n" + out[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
def test_reference_mpi_unittest_on_acc_backend(nranks, tmp_path):
    exe = os.path.join(ROOT, "oracle", "_ref", "host_acc_mpi", "dbcsr_unittest3")
    if not (os.path.exists(exe) and os.path.exists(MPIEXEC)):
        pytest.skip("host_acc_mpi or mpiexec not available")
    r = subprocess.run([MPIEXEC, "-n", str(nranks), exe], cwd=tmp_path, env=ENV, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " FAILED !" not in r.stdout.upper() and r.stdout.upper().count("PASSED !") > 0, r.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("name", ["test_square_sparse.perf", "test_square_dense.perf", "test_rect1_sparse.perf", "test_rect1_dense.perf", "test_H2O.perf"])
def test_resident_engine_under_a_multi_rank_fortran_host(name, nranks, tmp_path):
    """dbcsr_amd/fortran/dbcsr_amd_resident.F, multi_rank_multiply: the patched reference host on several MPI ranks -- each rank
    gathers the row panel of A over its process row and the column panel of B over its process column (MPI), multiplies them on
    the device into its part of C and puts that part back.  The driver's golden checksums must come out, and every rank must have
    taken the device path."""
    if GOLD[name]["check"] != "T":
        pytest.skip("no golden checksum in this input")
    out = run_perf("host_resident_mpi", name, nranks, tmp_path, dict(ENV, DBCSR_AMD_RESIDENT="1v"))
    ranks = set(int(m) for m in re.findall(r"dbcsr_amd_resident: rank\s+(\d+) of", out))
    assert ranks == set(range(nranks)), "not every rank multiplied on the device:\n" + out[-2500:]
    mm = re.search(r"matmuls total\s+(\d+)", out)
    assert mm and int(mm.group(1)) == 0, out[-3000:]   # no parameter stack was built anywhere


@pytest.mark.gpu
def test_resident_multi_rank_falls_through_together(tmp_path):
    """the reference's multiply unit tests (limits, symmetries, transposes, types) on two ranks of the patched host: whatever the
    multi-rank device path does not take must fall through to the reference path on ALL ranks together (a rank alone in a
    collective would hang), and every case must still pass"""
    exe = os.path.join(ROOT, "oracle", "_ref", "host_resident_mpi", "dbcsr_unittest3")
    if not (os.path.exists(exe) and os.path.exists(MPIEXEC)):
        pytest.skip("host_resident_mpi or mpiexec not available")
    r = subprocess.run([MPIEXEC, "-n", "2", exe], cwd=tmp_path, env=dict(ENV, DBCSR_AMD_RESIDENT="1"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " FAILED !" not in r.stdout.upper() and r.stdout.upper().count("PASSED !") > 0, r.stdout[-3000:]


def run_dump_and_compare_with_fixture(host, name, nranks, tmp_path, env):
    """dbcsr_ref_dump of `host` on `nranks` ranks (one output file per rank: its blocks, the global checksum); the union of the ranks'
    C blocks must be the block set of the reference's single-rank result (tests/golden/ref_dump.json), values to 1e-10"""
    import base64
    import sys

    import numpy as np
    from tests import ref_dump_util as R
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ref_fixtures as F
    exe = os.path.join(ROOT, "oracle", "_ref", host, "dbcsr_ref_dump")
    if not (os.path.exists(exe) and os.path.exists(MPIEXEC)):
        pytest.skip("%s or mpiexec not available" % host)
    ref = R.RefResult(name)
    F.write_nml(ref.params, str(tmp_path / "case.nml"))
    r = subprocess.run([MPIEXEC, "-n", str(nranks), exe, str(tmp_path / "case.nml"), str(tmp_path / "out.txt")], cwd=tmp_path,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    got = {}
    for q in range(nranks):
        d = F.parse_dump(str(tmp_path / ("out.txt.rank%d" % q)))
        assert abs(d["checksum"][0] / ref.checksum - 1.0) <= 1e-11   # (the checksum is global: every rank wrote the same one)
        vals = np.frombuffer(base64.b64decode(d["values_b64"]), "<f8") if d["nblks"] else np.zeros(0)
        off = 0
        for row, col, m, n in zip(d["row"], d["col"], d["m"], d["n"]):
            assert (row, col) not in got, "a block of C on two ranks"
            got[(row, col)] = vals[off:off + m * n]
            off += m * n
    assert len(got) == ref.nblks
    scale = max(float(np.max(np.abs(ref.data))), 1e-300) if ref.nblks else 1.0
    off = 0
    for b in range(ref.nblks):
        key = (int(ref.rows[b]) + 1, int(ref.col_i[b]) + 1)
        ne = int(ref.m[b] * ref.n[b])
        assert key in got, key
        assert np.max(np.abs(got[key] - ref.data[off:off + ne])) <= 1e-10 * scale, key
        off += ne
    return r.stdout


FILTER_CASES = ["filter_eps_mid", "filter_eps_mixed", "filter_eps_retain", "basic_5"]


@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("name", FILTER_CASES[:2])
def test_reference_mpi_filtered_multiply_equals_the_single_rank_fixture(name, nranks, tmp_path):
    """what the GPU test below relies on: the reference itself, on several ranks, filters to the block set of its single-rank run"""
    run_dump_and_compare_with_fixture("host_cpu_mpi", name, nranks, tmp_path, ENV)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("name", FILTER_CASES)
def test_resident_multi_rank_filtered_multiply_matches_the_reference(name, nranks, tmp_path):
    """filter_eps under the multi-rank glue: every rank holds the whole block row panel of A, so the on-the-fly filter sees the row
    counts the reference sums over the process row (dbcsr_mm_cannon.F:1040-1113) and takes the decisions one rank would."""
    out = run_dump_and_compare_with_fixture("host_resident_mpi", name, nranks, tmp_path, dict(ENV, DBCSR_AMD_RESIDENT="1v"))
    ranks = set(int(m) for m in re.findall(r"dbcsr_amd_resident: rank\s+(\d+) of", out))
    assert ranks == set(range(nranks)), "not every rank multiplied on the device:\n" + out[-2500:]


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("args", [("2316", "0.8", "23", "5"), ("1500", "0.5", "32", "4")], ids=["23", "32"])
def test_matrices_stay_on_the_device_under_a_multi_rank_fortran_host(args, nranks, tmp_path):
    """tests/fortran/dbcsr_resident_loop.F90 on several MPI ranks (sharing the box's GPU): dbcsr_amd_dev_create gathers the row panel of
    A and the column panel of B ONCE (MPI) and keeps them in HBM with this rank's tile of C, dbcsr_amd_dev_multiply then multiplies
    nrep times without moving anything between the ranks, dbcsr_amd_dev_download brings every rank's tile back once.  The checksums of
    the build's own dbcsr_multiply loop must come out (1e-10), on every grid MPI_Dims_create picks."""
    exe = os.path.join(ROOT, "oracle", "_ref", "host_resident_mpi", "dbcsr_resident_loop")
    if not (os.path.exists(exe) and os.path.exists(MPIEXEC)):
        pytest.skip("host_resident_mpi or mpiexec not available")
    r = subprocess.run([MPIEXEC, "-n", str(nranks), exe, *args, "1"], cwd=tmp_path, env=dict(ENV, DBCSR_AMD_RESIDENT="0"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert re.search(r"ranks\s+%d\b" % nranks, r.stdout), r.stdout[-2000:]
    d = re.search(r"relative difference\s+([0-9.E+-]+)", r.stdout)
    assert d and float(d.group(1)) <= 1e-10, r.stdout[-2000:]
    assert re.search(r"steady state \(third multiply on\) per multiply \[s\]\s+[0-9.]+", r.stdout), r.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("mode", ["1", "2"])
def test_resident_products_out_of_place_and_as_operands_on_several_ranks(mode, nranks, tmp_path):
    """round 5: a purification-style loop on several ranks.  Mode 1: C_in = the original C in every multiply, the product goes to a second
    resident matrix; mode 2: X_{n+1} = beta C + alpha X_n B -- the product (this rank's TILE) becomes the left operand of the next multiply
    through dbcsr_amd_dev_as_operand, which gathers its row panel from the peers' tiles (make_m2s' job, src/mm/dbcsr_mm_cannon.F:146-258)
    without a download / dbcsr_type / create round trip.  The checksums of the same loop through dbcsr_multiply must come out (1e-10)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "host_resident_mpi", "dbcsr_resident_loop")
    if not (os.path.exists(exe) and os.path.exists(MPIEXEC)):
        pytest.skip("host_resident_mpi or mpiexec not available")
    r = subprocess.run([MPIEXEC, "-n", str(nranks), exe, "2316", "0.8", "23", "4", "1", mode], cwd=tmp_path, env=dict(ENV, DBCSR_AMD_RESIDENT="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert re.search(r"ranks\s+%d\b" % nranks, r.stdout) and "resident_loop: mode %s" % mode in r.stdout, r.stdout[-2000:]
    d = re.search(r"relative difference\s+([0-9.E+-]+)", r.stdout)
    assert d and float(d.group(1)) <= 1e-10, r.stdout[-2000:]


@pytest.mark.gpu
def test_resident_rccl_request_falls_back_when_ranks_share_a_device(tmp_path):
    """DBCSR_AMD_RESIDENT_RCCL=1 with two ranks on the box's one GPU: RCCL cannot serve two ranks of a communicator on one device; the
    glue must notice on every rank (fewer devices than ranks), agree, and move the panels through MPI -- same results, no hang"""
    out = run_perf("host_resident_mpi", "test_square_sparse.perf", 2, tmp_path, dict(ENV, DBCSR_AMD_RESIDENT="1v", DBCSR_AMD_RESIDENT_RCCL="1"))
    assert "panels travel over RCCL" not in out
    ranks = set(int(m) for m in re.findall(r"dbcsr_amd_resident: rank\s+(\d+) of", out))
    assert ranks == {0, 1}, out[-2500:]


# cases of tests/golden/ref_dump.json with op(), symmetric / antisymmetric operands or limits (product matrices without symmetry)
OP_CASES = ["alpha_beta_mixed_TN", "mixed_NT", "mixed_TT", "symm_a_S", "symm_a_A", "symm_a_S_T", "symm_ab_S", "symm_b_A_T", "symm_b_S",
            "limits_T", "limits_beta0", "limits_beta0_k", "limits_cut_new", "limits_retain"]


@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("name", ["mixed_TT", "symm_a_A", "limits_T", "symm_c_S_NT", "symm_c_A_NT"])
def test_reference_mpi_op_symmetry_limits_equal_the_single_rank_fixture(name, nranks, tmp_path):
    """what the GPU test below relies on: the reference itself, on several ranks, gives the block set and the values of its single-rank run"""
    run_dump_and_compare_with_fixture("host_cpu_mpi", name, nranks, tmp_path, ENV)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("name", OP_CASES)
def test_resident_multi_rank_op_symmetry_and_limits_on_the_device(name, nranks, tmp_path):
    """Transposed, symmetric / antisymmetric operands and submatrix limits under a multi-rank Fortran host (round 3: they fell through to
    the reference's CPU path): every rank sends its blocks of op(desym(A)) / op(desym(B)) to the process row / column that needs them
    (gather_panel_general: make_m2s' redistribution, src/mm/dbcsr_mm_cannon.F:146-258), the device multiplies the panels and crops by the
    limits as under one rank.  Block set and values of the reference's single-rank result, every rank on the device path."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from tests import ref_dump_util as R
    out = run_dump_and_compare_with_fixture("host_resident_mpi", name, nranks, tmp_path, dict(ENV, DBCSR_AMD_RESIDENT="1v"))
    ranks = set(int(m) for m in re.findall(r"dbcsr_amd_resident: rank\s+(\d+) of", out))
    assert ranks == set(range(nranks)), "not every rank multiplied on the device:\n" + out[-2500:]
    p = R.RefResult(name).params
    if p["transa"] != "N" or p["transb"] != "N" or p["symm_a"] != "N" or p["symm_b"] != "N":
        assert "general gather" in out, out[-2500:]
    if any(p["limits"]):
        assert "[limits]" in out, out[-2500:]


# round 6: product matrices WITH symmetry on several ranks (VERDICT r05 missing 4; src/mm/dbcsr_mm.F:529-575, 711-719)
SYMC_CASES = ["symm_c_S_NT", "symm_c_S_TN", "symm_c_S_NN", "symm_c_S_beta0", "symm_c_S_retain", "symm_c_S_filter", "symm_c_A_NT", "symm_abc_S"]


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("name", SYMC_CASES)
def test_resident_multi_rank_symmetric_product_on_the_device(name, nranks, tmp_path):
    """A symmetric / antisymmetric PRODUCT matrix under a multi-rank Fortran host: a rank holds the blocks of the stored triangle whose canonical
    position (checker_tr) lies in its tile, some stored transposed; they go up turned back, dbcsr_amd_multiply_symmetric_c computes the
    canonical blocks of the panels' product, and every rank puts back exactly the blocks the reference keeps there.  Block set (no block on
    two ranks, none missing) and values of the reference's single-rank result; every rank on the device path."""
    out = run_dump_and_compare_with_fixture("host_resident_mpi", name, nranks, tmp_path, dict(ENV, DBCSR_AMD_RESIDENT="1v"))
    ranks = set(int(m) for m in re.findall(r"dbcsr_amd_resident: rank\s+(\d+) of", out))
    assert ranks == set(range(nranks)), "not every rank multiplied on the device:\n" + out[-2500:]
    assert "[product with symmetry]" in out, out[-2500:]


@pytest.mark.gpu
def test_reference_unit_tests_take_the_device_path_on_two_ranks(tmp_path):
    """the reference's multiply unit tests (tests/dbcsr_unittest1.F: every combination of transposes, symmetries, limits, retain_sparsity,
    types) on two ranks of the patched host: all pass, and the multiplies with op() / symmetric operands / limits are counted on the
    device path (what is left to the reference path: complex data, product matrices with symmetry)"""
    exe = os.path.join(ROOT, "oracle", "_ref", "host_resident_mpi", "dbcsr_unittest1")
    if not (os.path.exists(exe) and os.path.exists(MPIEXEC)):
        pytest.skip("host_resident_mpi or mpiexec not available")
    r = subprocess.run([MPIEXEC, "-n", "2", exe], cwd=tmp_path, env=dict(ENV, DBCSR_AMD_RESIDENT="1v"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " FAILED !" not in r.stdout.upper() and r.stdout.upper().count("PASSED !") > 0, r.stdout[-3000:]
    dev = len(re.findall(r"dbcsr_amd_resident: rank\s+0 of", r.stdout))
    general = len(re.findall(r"dbcsr_amd_resident: rank\s+0 of.*general gather", r.stdout))
    limited = len(re.findall(r"dbcsr_amd_resident: rank\s+0 of.*\[limits\]", r.stdout))
    symc = len(re.findall(r"dbcsr_amd_resident: rank\s+0 of.*\[product with symmetry\]", r.stdout))
    back = re.findall(r"dbcsr_amd_resident: left to the reference path:(.*)", r.stdout)
    print("dbcsr_unittest1 on 2 ranks: %d multiplies on the device, %d of them through the general gather, %d with limits, %d with a symmetric product; "
          "left to the reference path: %s" % (dev, general, limited, symc, sorted(set(b.strip() for b in back))))
    assert dev > 20 and general > 10 and limited > 0 and symc > 0, (dev, general, limited, symc)
