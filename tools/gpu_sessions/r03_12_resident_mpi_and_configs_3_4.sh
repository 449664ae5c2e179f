#!/bin/bash
# round 3, GPU session 12: the resident engine under a multi-rank Fortran host; configs 3 / 4 with the 256 MB panel default
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s12; mkdir -p $O
timeout 1800 python -m pytest tests/test_fortran_host_mpi.py -q -m gpu -k "resident" > $O/pytest_resident_mpi.txt 2>&1
tail -25 $O/pytest_resident_mpi.txt
for wl in config3_32768_mixed13_23_32_fill5_fp64 config4_131072_23x23_fill1_fp64 config1_4096_4x4_fill10_fp64; do for mb in 160 256; do
  DBCSR_AMD_MM_PANEL_MB=$mb timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --no-pmc --cpu-seconds 0 > $O/b_${wl}_$mb.json 2> $O/b_${wl}_$mb.err
  python - $O/b_${wl}_$mb.json $wl $mb <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("%s panel %s MB: %.3f ms/step  kernel %.3f ms  %.1f GFLOP/s  %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], r["kernel_ms"], d["value"], r["kernel"][:40]))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done; done
