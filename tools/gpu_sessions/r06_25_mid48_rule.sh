#!/bin/bash
# round 6, GPU session 25: parity after the rule for 41 ... 48 (slab kernel when the other dimension is at most 40 or both are multiples of 4)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s25; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 900 python -m pytest tests/test_gpu_big_blocks.py tests/test_gpu_kernel_variants.py -q -x 2>&1 | grep -v "$F" | tail -8 ) > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
DBCSR_AMD_SWEEP_FORCED=60 DBCSR_AMD_SWEEP_PLAIN=160 DBCSR_AMD_SWEEP_BIG=200 DBCSR_AMD_SWEEP_MID=600 timeout 900 python -m pytest tests/test_gpu_random_sweep.py -q -x -n 4 2>&1 | grep -v "$F" | tail -3
