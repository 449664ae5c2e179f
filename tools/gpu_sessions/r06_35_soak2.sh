#!/bin/bash
# round 6, GPU session 35: a second, longer soak on the last tree with cases no earlier walk has seen (DBCSR_AMD_SWEEP_OFFSET shifts every generator seed):
# 3000 forced-kernel + 8000 plain + 3000 large-block + 4000 cases with blocks of 33 ... 40 against the oracle, 1500 random parameter stacks, 500 exact-size stack kernels
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s35; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
export DBCSR_AMD_SWEEP_OFFSET=700000
( time DBCSR_AMD_SWEEP_FORCED=3000 DBCSR_AMD_SWEEP_PLAIN=8000 DBCSR_AMD_SWEEP_BIG=3000 DBCSR_AMD_SWEEP_MID=4000 timeout 3000 python -m pytest tests/test_gpu_random_sweep.py -q -x -n 6 2>&1 | grep -v "$F" | tail -8 ) > $O/sweep.txt 2>&1
tail -6 $O/sweep.txt
( time DBCSR_AMD_SWEEP_STACKS=1500 DBCSR_AMD_SWEEP_EXACT_STACKS=500 timeout 2400 python -m pytest tests/test_gpu_libsmm.py tests/test_gpu_smm_exact.py -q -x -n 4 2>&1 | grep -v "$F" | tail -12 ) > $O/stacks.txt 2>&1
tail -5 $O/stacks.txt
