#!/bin/bash
# round 6, GPU session 9: sub-blocks in units of 4 x 4 in the workgroup kernels of the blocks of 33 ... 80 (BigSub):
# parity (engine + acc ABI + randomised sweep over large blocks), then the block_bench / acc_bench lines of profiles/r05_large_blocks.txt again
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s09; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 900 python -m pytest tests/test_gpu_big_blocks.py tests/test_gpu_libsmm.py tests/test_gpu_smm_exact.py -q -x 2>&1 | grep -v "$F" | tail -12 ) > $O/pytest_big.txt 2>&1
tail -6 $O/pytest_big.txt
DBCSR_AMD_SWEEP_BIG=120 timeout 400 python -m pytest tests/test_gpu_random_sweep.py -q -k "large_blocks" 2>&1 | grep -v "$F" | tail -6 > $O/pytest_big_sweep.txt
tail -3 $O/pytest_big_sweep.txt
B='[{"mix":"1,33","fill":0.2},{"mix":"1,36","fill":0.2},{"mix":"1,40","fill":0.2},{"mix":"1,55","fill":0.3},{"mix":"1,64","fill":0.3},{"mix":"1,72","fill":0.3},{"mix":"1,80","fill":0.3},'
B="$B"'{"mix_m":"1,45","mix_n":"1,67","mix_k":"1,78","fill":0.3},{"mix":"1,49","fill":0.3},{"mix":"1,69","fill":0.3}]'
timeout 400 python tools/block_bench.py --size 16384 --label r06_big_sub4 --check --batch "$B" 2>&1 | grep -v "$F" > $O/large_blocks.jsonl
cut -c1-330 $O/large_blocks.jsonl
for mnk in "33 33 33" "36 36 36" "40 40 40" "55 55 55" "64 64 64" "72 72 72" "80 80 80" "45 67 78"; do
  timeout 120 python tools/acc_bench.py 5 16005 $mnk --check 2>&1 | grep acc_bench >> $O/acc_bench_blocks.txt
done
cut -c1-200 $O/acc_bench_blocks.txt
