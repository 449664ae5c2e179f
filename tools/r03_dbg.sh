#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r03_dbg; mkdir -p $O
python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
from tests.test_gpu_fortran_host import write_perf
G = json.load(open("tests/golden/perf_golden.json"))
c = dict(G["test_square_sparse.perf"]); c["npcols"] = 0
write_perf(c, "gpurun_out/r03_dbg/case.perf")
PY
cd $O && DBCSR_AMD_RESIDENT=1v OMP_NUM_THREADS=2 MKL_THREADING_LAYER=SEQUENTIAL timeout 300 /opt/conda/bin/mpiexec -n 2 $GRAFT_REPO_ROOT/oracle/_ref/host_resident_mpi/dbcsr_perf case.perf > out.txt 2>&1
echo rc $?
grep -v "^ DBCSR|\|^$" out.txt | head -80
