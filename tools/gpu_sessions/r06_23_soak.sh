#!/bin/bash
# round 6, GPU session 23: a longer randomised soak of the multiply against the oracle after the round's kernel changes (padded pitches, slab kernels,
# exact-size stack kernels): 1200 mixed cases, 400 large-block cases, 600 cases with blocks of 33 ... 40, 300 random parameter stacks
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s23; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time DBCSR_AMD_SWEEP_FORCED=400 DBCSR_AMD_SWEEP_PLAIN=1200 DBCSR_AMD_SWEEP_BIG=400 DBCSR_AMD_SWEEP_MID=600 timeout 2400 python -m pytest tests/test_gpu_random_sweep.py -q -x -n 4 2>&1 | grep -v "$F" | tail -6 ) > $O/sweep.txt 2>&1
tail -5 $O/sweep.txt
( time DBCSR_AMD_SWEEP_STACKS=300 DBCSR_AMD_SWEEP_EXACT_STACKS=200 timeout 1500 python -m pytest tests/test_gpu_libsmm.py tests/test_gpu_smm_exact.py -q -x 2>&1 | grep -v "$F" | tail -12 ) > $O/stacks.txt 2>&1
tail -4 $O/stacks.txt
