#!/bin/bash
# round 4, GPU session 18: a long walk of the randomised parity sweep (1200 plain + 480 forced-path cases instead of 160 + 60)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s18; mkdir -p $O
DBCSR_AMD_SWEEP_PLAIN=1200 DBCSR_AMD_SWEEP_FORCED=480 timeout 1300 python -m pytest tests/test_gpu_random_sweep.py -q -n 4 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/soak.txt; tail -25 $O/soak.txt
