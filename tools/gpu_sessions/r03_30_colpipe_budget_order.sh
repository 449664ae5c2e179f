#!/bin/bash
# round 3, GPU session 30: does the 8-rank row of the colpipe step budget depend on what ran before it in the same process?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s30; mkdir -p $O
( for r in 8 4,8 2,8 2,4,8; do for steps in 6 10; do
  timeout 900 python tools/rank_step_budget.py --ranks $r --steps $steps --colpipe 8 | sed "s/^# workload/# ranks $r, steps $steps; workload/" | grep -v "^# ranks grid"; done; done ) 2>&1 | grep -v amdgpu.ids | tee $O/order.txt
