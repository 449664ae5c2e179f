#!/bin/bash
# round 5, GPU session 13: stack entries per workgroup of the acc ABI's large-block kernel (flushes with atomic adds per run end)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s13; mkdir -p $O
for g in 8 4 16 32; do
  for mnk in "72 72 72" "40 40 40" "64 64 64"; do
    DBCSR_AMD_SMM_BIG_GROUP=$g timeout 120 python tools/acc_bench.py 5 16005 $mnk 2>&1 | grep acc_bench | sed "s/^/[group $g] /" >> $O/acc_bench_group.txt
  done
done
cat $O/acc_bench_group.txt
