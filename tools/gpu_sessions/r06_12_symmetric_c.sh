#!/bin/bash
# round 6, GPU session 12: product matrices with symmetry on several ranks of the patched Fortran host, blocks stored transposed
# turned back on the way up, the counter of multiplies left to the reference path; the Fortran-host suites around them
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s12; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 1500 python -m pytest tests/test_fortran_host_mpi.py -m gpu -q -x -s -k "symmetric_product or unit_tests_take or falls_through or op_symmetry" 2>&1 | grep -v "$F" | tail -60 ) > $O/pytest_mpi.txt 2>&1
grep "multiplies on the device" $O/pytest_mpi.txt | grep -v print; tail -6 $O/pytest_mpi.txt
( time timeout 1200 python -m pytest tests/test_gpu_fortran_host.py -q -x 2>&1 | grep -v "$F" | tail -8 ) > $O/pytest_serial.txt 2>&1
tail -5 $O/pytest_serial.txt
