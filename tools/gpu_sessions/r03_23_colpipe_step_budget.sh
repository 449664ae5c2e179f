#!/bin/bash
# round 3, GPU session 23: compute path of one rank of the colpipe schedule (N x 1 grid, column chunks), alone on the GPU, against the gather schedule's
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s23; mkdir -p $O
( timeout 900 python tools/rank_step_budget.py --ranks 2,4,8
  for nc in 2 4 8; do timeout 900 python tools/rank_step_budget.py --ranks 2,4,8 --colpipe $nc | sed "s/^# workload/# colpipe, $nc column chunks; workload/"; done ) 2>&1 | grep -v amdgpu.ids | tee $O/rank_step_budget_colpipe.txt
