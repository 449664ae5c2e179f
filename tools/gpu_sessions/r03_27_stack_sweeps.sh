#!/bin/bash
# round 3, GPU session 27: acc-ABI stack kernels -- inhomogeneous stacks of several lengths, and the random (m, n, k) / stack sweep that is off by default
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s27; mkdir -p $O
DBCSR_AMD_SWEEP_STACKS=120 timeout 1200 python -m pytest tests/test_gpu_libsmm.py -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
