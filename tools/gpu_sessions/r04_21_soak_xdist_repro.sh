#!/bin/bash
# round 4, GPU session 21: the two failures of session 18 happened with four pytest workers sharing the GPU and not in one process:
# the same stretch of the walk (plain cases 640 .. 899) with four workers again, this time with the report
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s21; mkdir -p $O
IDS=""
for s in $(seq 640 899); do IDS="$IDS tests/test_gpu_random_sweep.py::test_random_multiply_matches_oracle[$s]"; done
DBCSR_AMD_SWEEP_PLAIN=1200 DBCSR_AMD_SWEEP_FORCED=0 timeout 420 python -m pytest $IDS -q -n 4 -rf --tb=short 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -150 > $O/xdist.txt; tail -100 $O/xdist.txt
