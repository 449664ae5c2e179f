"""The UNCHANGED reference (DBCSR library + its performance driver, expanded with tools/fypp_lite.py, compiled with
amdflang, BLAS = MKL; tools/build_dbcsr_host.py cpu) on its own golden .perf inputs, here in the build container:
proves that the expander + build reproduce the reference (the driver checks its checksums itself at 1e-11) -- the
fixtures of tests/golden/ref_dump.json come from this very build."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "host_cpu", "dbcsr_perf")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "perf_golden.json")))

pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="reference Fortran host not built (tools/build_dbcsr_host.py cpu)")


@pytest.mark.parametrize("name", sorted(k for k, v in GOLD.items() if v["check"] == "T"))
def test_reference_cpu_host_reproduces_its_golden_checksums(name, tmp_path):
    from tests.test_gpu_fortran_host import write_perf
    write_perf(GOLD[name], tmp_path / "case.perf")
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="4")
    r = subprocess.run([EXE, str(tmp_path / "case.perf")], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "checksum(C_out)" in r.stdout
