#!/bin/bash
# round 6, GPU session 21: one rank's share of an N-GPU step, ALONE on the device (tools/rank_step_budget.py), with the round's kernels: the gather schedule's
# one multiply, the colpipe schedule on the N x 1 grid and on the 2-D grid (colpipe2d), 4 and 8 column chunks -- the inputs of tools/schedule_model.py
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s21; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for W in config2_32768_23x23_fill10_fp64 config4_131072_23x23_fill1_fp64; do
  timeout 600 python tools/rank_step_budget.py --workload $W --ranks 1,2,4,8 2>&1 | grep -v "$F" >> $O/budget.txt
  for C in 4 8; do
    echo "# colpipe (N x 1 grid), $C column chunks" >> $O/budget.txt
    timeout 600 python tools/rank_step_budget.py --workload $W --ranks 2,4,8 --colpipe $C 2>&1 | grep -v "$F" >> $O/budget.txt
    echo "# colpipe2d (2-D grid), $C column chunks" >> $O/budget.txt
    timeout 600 python tools/rank_step_budget.py --workload $W --ranks 4,8 --colpipe2d $C 2>&1 | grep -v "$F" >> $O/budget.txt
  done
done
cat $O/budget.txt
