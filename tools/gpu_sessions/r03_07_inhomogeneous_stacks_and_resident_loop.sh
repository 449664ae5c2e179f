#!/bin/bash
# round 3, GPU session 7: inhomogeneous stacks on the device (unchanged host incl. G2G), resident Fortran loop
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s07; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_libsmm.py tests/test_gpu_fortran_host.py tests/test_gpu_acc_spec.py -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
cd /tmp && for args in "8192 0.9 23 6 1" "32768 0.9 23 8 0"; do
  echo "== dbcsr_resident_loop $args"; OMP_NUM_THREADS=8 MKL_THREADING_LAYER=SEQUENTIAL timeout 900 $GRAFT_REPO_ROOT/oracle/_ref/host_resident/dbcsr_resident_loop $args 2>&1 | grep "resident_loop"
done > $GRAFT_REPO_ROOT/$O/resident_loop.txt 2>&1
cat $GRAFT_REPO_ROOT/$O/resident_loop.txt
