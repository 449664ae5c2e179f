#!/bin/bash
# round 6, GPU session 6: the same with operands that fit one L2 (400 A and 400 B blocks instead of 10000 each): what the stack kernels do when
# the fabric is out of the way -- the benchmark's own shape moves two random operand blocks per product over it
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s06; mkdir -p $O
for mnk in "23 23 23" "13 13 13" "32 32 32" "4 4 4"; do
  for T in 1 16; do
    timeout 200 python tools/acc_bench.py 20 30000 $mnk 2000 400 400 --threads $T --check 2>&1 | grep acc_bench >> $O/acc_threads_l2.txt
  done
done
cat $O/acc_threads_l2.txt
