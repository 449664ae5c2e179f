#!/bin/bash
# round 4, GPU session 11: op() / symmetric operands / limits under a multi-rank Fortran host on the device (general gather)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s11; mkdir -p $O
timeout 1500 python -m pytest tests/test_fortran_host_mpi.py -k "op_symmetry_and_limits_on_the_device or unit_tests_take_the_device_path or falls_through or resident_engine" -x -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -25 | tee $O/pytest.txt
