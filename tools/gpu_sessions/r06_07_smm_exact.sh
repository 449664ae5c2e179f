#!/bin/bash
# round 6, GPU session 7: the exact-size stack kernel (smm_exact.h, hiprtc per triplet): parity, the reference-host tests that run through the
# acc ABI, then the threaded acc_bench lines of sessions 5 / 6 again
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s07; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_smm_exact.py tests/test_gpu_libsmm.py tests/test_gpu_acc_spec.py -q -x 2>&1 | grep -v "$F" | tail -8 > $O/pytest_smm_exact.txt
tail -5 $O/pytest_smm_exact.txt
for mnk in "23 23 23" "13 13 13" "32 32 32" "4 4 4" "5 5 5" "13 23 32"; do
  for T in 1 16; do
    timeout 200 python tools/acc_bench.py 20 30000 $mnk 2000 --threads $T --check 2>&1 | grep acc_bench >> $O/acc_threads.txt
    timeout 200 python tools/acc_bench.py 20 30000 $mnk 2000 400 400 --threads $T --check 2>&1 | grep acc_bench | sed 's/$/  [400 A, 400 B blocks]/' >> $O/acc_threads.txt
  done
done
for mnk in "23 23 23" "13 13 13" "32 32 32"; do
  DBCSR_AMD_SMM_EXACT=0 timeout 200 python tools/acc_bench.py 20 30000 $mnk 2000 400 400 --threads 16 2>&1 | grep acc_bench | sed 's/$/  [400 A, 400 B blocks; DBCSR_AMD_SMM_EXACT=0]/' >> $O/acc_threads.txt
done
timeout 100 python tools/acc_bench.py 20 16005 23 23 23 --threads 1 --check 2>&1 | grep acc_bench >> $O/acc_threads.txt
cat $O/acc_threads.txt
