#!/bin/bash
# round 5, GPU session 14: the randomised sweep over matrices with blocks of 33 .. 80 (60 + 240 further seeds)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s14; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
DBCSR_AMD_SWEEP_BIG=300 timeout 400 python -m pytest tests/test_gpu_random_sweep.py -q -k "large_blocks" 2>&1 | grep -v "$F" | tail -25 > $O/pytest_big_sweep.txt
tail -12 $O/pytest_big_sweep.txt
