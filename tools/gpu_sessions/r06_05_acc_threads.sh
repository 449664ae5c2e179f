#!/bin/bash
# round 6, GPU session 5: the acc ABI with the concurrency of the real host -- T host threads, each with its own stream,
# 30000-entry stack (mm_stack_size) and C blocks, calling libsmm_acc_process at once (tools/acc_bench.py --threads)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s05; mkdir -p $O
for mnk in "23 23 23" "4 4 4" "13 13 13" "32 32 32"; do
  for T in 1 4 16; do
    timeout 200 python tools/acc_bench.py 20 30000 $mnk 2000 --threads $T --check 2>&1 | grep acc_bench >> $O/acc_threads.txt
  done
done
timeout 100 python tools/acc_bench.py 20 16005 23 23 23 --threads 1 --check 2>&1 | grep acc_bench >> $O/acc_threads.txt
cat $O/acc_threads.txt
