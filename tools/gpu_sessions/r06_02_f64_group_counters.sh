#!/bin/bash
# round 6, GPU session 2: counters of the fp64 group kernel (R = 4, R = 2) and of the production kernel on the same box, config 2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s02; mkdir -p $O
for spec in "group_R4:DBCSR_AMD_MM_F64_GROUP=4" "group_R2:DBCSR_AMD_MM_F64_GROUP=2" "hot:DBCSR_AMD_MM_F64_GROUP=0"; do
  L=${spec%%:*}; E=${spec#*:}
  ( export $E; bash tools/profile_cmd.sh r06_$L python $PWD/tools/block_bench.py --size 32768 --mix 1,23 --fill 0.1 --steps 2 > $O/prof_$L.txt 2>&1 )
  cp gpurun_out/prof_r06_$L/summary.txt $O/summary_$L.txt
done
tail -80 $O/summary_group_R4.txt
