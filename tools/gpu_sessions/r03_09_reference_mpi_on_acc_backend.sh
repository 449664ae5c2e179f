#!/bin/bash
# round 3, GPU session 9: the reference as an MPI program on this back end (ranks share the GPU); config 2 through it
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s09; mkdir -p $O
which mpiexec; nproc
timeout 1800 python -m pytest tests/test_fortran_host_mpi.py -q -m gpu > $O/pytest_mpi.txt 2>&1
tail -15 $O/pytest_mpi.txt
timeout 2400 bash tools/refdriver_config2_mpi.sh > $O/refdriver_config2_mpi.txt 2>&1
cat $O/refdriver_config2_mpi.txt
