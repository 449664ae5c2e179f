#!/bin/bash
# round 3, GPU session 25: filter_eps under the multi-rank Fortran glue (resident_mpi, ranks sharing the GPU)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s25; mkdir -p $O
timeout 1500 python -m pytest tests/test_fortran_host_mpi.py -q -m gpu -k "resident" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
