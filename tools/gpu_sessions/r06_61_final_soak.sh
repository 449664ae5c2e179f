#!/bin/bash
# round 6, GPU session 61: the last tree once more against the oracle with cases no earlier walk has seen (offset 900000), the kernel families and 24 class shapes at scale
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s61; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
export DBCSR_AMD_SWEEP_OFFSET=900000
( time DBCSR_AMD_SWEEP_FORCED=2000 DBCSR_AMD_SWEEP_PLAIN=6000 DBCSR_AMD_SWEEP_BIG=2000 DBCSR_AMD_SWEEP_MID=2000 timeout 3000 python -m pytest tests/test_gpu_random_sweep.py -q -x -n 6 2>&1 | grep -v "$F" | tail -4 ) > $O/sweep.txt 2>&1
tail -4 $O/sweep.txt
( time DBCSR_AMD_SWEEP_STACKS=1000 DBCSR_AMD_SWEEP_EXACT_STACKS=300 timeout 2400 python -m pytest tests/test_gpu_libsmm.py tests/test_gpu_smm_exact.py -q -x -n 4 2>&1 | grep -v "$F" | tail -4 ) > $O/stacks.txt 2>&1
tail -3 $O/stacks.txt
( time DBCSR_AMD_CLASS_SHAPES=24 timeout 2400 python -m pytest tests/test_gpu_class_mode.py -q -x 2>&1 | grep -v "$F" | tail -4 ) > $O/scale.txt 2>&1
tail -3 $O/scale.txt
