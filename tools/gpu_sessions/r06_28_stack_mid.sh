#!/bin/bash
# round 6, GPU session 28: blocks of 33 ... 40 under the acc ABI through the one-wave slab kernel (smm_stack_f64_mid) against the workgroup kernel
# (DBCSR_AMD_SMM_MID=0): parity, acc_bench with one stream (the reference timer's stack) and sixteen
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s28; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( DBCSR_AMD_SWEEP_STACKS=200 timeout 900 python -m pytest tests/test_gpu_libsmm.py tests/test_gpu_smm_exact.py tests/test_gpu_acc_spec.py -q -x 2>&1 | grep -v "$F" | tail -5 ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for mnk in "33 33 33" "36 36 36" "40 40 40" "37 34 40"; do
  for M in 1 0; do
    DBCSR_AMD_SMM_MID=$M timeout 120 python tools/acc_bench.py 5 16005 $mnk --check 2>&1 | grep acc_bench | cut -c1-200 >> $O/acc_bench.txt
    DBCSR_AMD_SMM_MID=$M timeout 200 python tools/acc_bench.py 10 30000 $mnk 2000 400 400 --threads 16 2>&1 | grep acc_bench | cut -c1-200 >> $O/acc_bench.txt
  done
done
cat $O/acc_bench.txt
