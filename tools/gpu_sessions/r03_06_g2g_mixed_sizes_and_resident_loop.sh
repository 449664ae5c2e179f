#!/bin/bash
# round 3, GPU session 6: G2G failure of the unchanged host on mixed block sizes; resident Fortran loop
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s06; mkdir -p $O
python - <<'PY' > $O/g2g_debug.txt 2>&1
import os, sys, subprocess, tempfile
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_ref_fixtures as F
from tests import ref_dump_util as R
exe = os.path.join("oracle", "_ref", "host_acc", "dbcsr_ref_dump")
for name in ("config3_like", "mixed_NT", "h2o_like_23", "filter_eps_mid"):
    for thr in ("1", "4"):
        ref = R.RefResult(name)
        with tempfile.TemporaryDirectory() as td:
            nml, out = os.path.join(td, "case.nml"), os.path.join(td, "out.txt")
            F.write_nml(ref.params, nml)
            e = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS=thr, DBCSR_USE_ACC_G2G="1")
            r = subprocess.run([os.path.abspath(exe), nml, out], cwd=td, env=e, capture_output=True, text=True, timeout=300)
            print("==", name, "threads", thr, "rc", r.returncode, "out exists", os.path.exists(out))
            if r.returncode != 0:
                print(r.stdout[-1500:]); print(r.stderr[-2500:])
PY
cat $O/g2g_debug.txt | head -120
cd /tmp && for args in "2316 0.8 23 4 1" "8192 0.9 23 4 1" "32768 0.9 23 6 0"; do
  echo "== dbcsr_resident_loop $args"; OMP_NUM_THREADS=8 MKL_THREADING_LAYER=SEQUENTIAL timeout 900 $GRAFT_REPO_ROOT/oracle/_ref/host_resident/dbcsr_resident_loop $args 2>&1 | grep -v "^ DBCSR\|^$" | tail -12
done > $GRAFT_REPO_ROOT/$O/resident_loop.txt 2>&1
cat $GRAFT_REPO_ROOT/$O/resident_loop.txt
