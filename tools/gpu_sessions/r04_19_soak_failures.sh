#!/bin/bash
# round 4, GPU session 19: session 18's long walk again in ONE process (four pytest workers on one GPU ran 10x slower and its time limit
# cut the report off after two failures): 1200 plain + 480 forced-path cases, with the failure report
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s19; mkdir -p $O
DBCSR_AMD_SWEEP_PLAIN=1200 DBCSR_AMD_SWEEP_FORCED=480 timeout 800 python -m pytest tests/test_gpu_random_sweep.py -q -rf --tb=short 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -150 > $O/failures.txt; tail -100 $O/failures.txt
