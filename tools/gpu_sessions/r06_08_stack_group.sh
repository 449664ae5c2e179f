#!/bin/bash
# round 6, GPU session 8: entries per wave of the exact stack kernel (DBCSR_AMD_STACK_GROUP) for one stream and for sixteen
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s08; mkdir -p $O
for mnk in "23 23 23" "32 32 32" "13 13 13"; do
  for G in 4 8 16 32 64; do
    for T in 1 16; do
      DBCSR_AMD_STACK_GROUP=$G timeout 200 python tools/acc_bench.py 20 30000 $mnk 2000 400 400 --threads $T 2>&1 | grep acc_bench | sed "s/^/group $G: /" >> $O/group_sweep.txt
    done
  done
done
for G in 8 16 32 64; do
  DBCSR_AMD_STACK_GROUP=$G timeout 200 python tools/acc_bench.py 20 30000 23 23 23 2000 --threads 16 2>&1 | grep acc_bench | sed "s/^/group $G: /" >> $O/group_sweep.txt
done
cut -c1-150 $O/group_sweep.txt
