#!/bin/bash
# acc-ABI stack kernel: GFLOP/s as a function of the stack entries per wavefront
for sz in "16005 23" "30000 23" "16005 13" "16005 32" "30000 5"; do set -- $sz
for g in 2 4 8 16; do
  echo "group=$g $(DBCSR_AMD_STACK_GROUP=$g python tools/acc_bench.py 20 $1 $2 2>&1 | tail -1)"
done; done
