#!/bin/bash
# round 5, GPU session 8: the large-block kernel under the acc ABI (smm_stack_f64_big) -- parity through the oracle, the reference's validator,
# the unchanged Fortran host's unit tests -- and its acc_bench numbers; the engine's large-block kernel after its LDS slab was trimmed
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s08; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 600 python -m pytest tests/test_gpu_libsmm.py tests/test_gpu_big_blocks.py tests/test_gpu_acc_spec.py tests/test_gpu_fortran_host.py tests/test_gpu_kernel_variants.py -q 2>&1 | grep -v "$F" | tail -15 ) > $O/pytest_abi_big.txt 2>&1
tail -6 $O/pytest_abi_big.txt
for mnk in "40 40 40" "72 72 72" "45 67 78" "64 64 64" "80 80 80" "33 33 33" "55 55 55" "23 23 23"; do
  timeout 120 python tools/acc_bench.py 5 16005 $mnk --check 2>&1 | grep acc_bench >> $O/acc_bench_blocks.txt
done
DBCSR_AMD_SMM_BIG=0 timeout 120 python tools/acc_bench.py 5 16005 72 72 72 2>&1 | grep acc_bench | sed 's/^/[DBCSR_AMD_SMM_BIG=0] /' >> $O/acc_bench_blocks.txt
cat $O/acc_bench_blocks.txt
B='[{"mix":"1,40","fill":0.2},{"mix":"1,64","fill":0.3},{"mix":"1,72","fill":0.3},{"mix":"1,80","fill":0.3},{"mix":"1,33","fill":0.2},{"mix":"1,55","fill":0.3},'
B="$B"'{"mix_m":"1,45","mix_n":"1,67","mix_k":"1,78","fill":0.3},{"mix":"1,72","fill":0.1,"size":32768}]'
timeout 300 python tools/block_bench.py --size 16384 --label r05_big_lds_trimmed --check --batch "$B" 2>&1 | grep -v "$F" > $O/large_blocks_final.jsonl
cut -c1-330 $O/large_blocks_final.jsonl
