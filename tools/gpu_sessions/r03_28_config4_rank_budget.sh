#!/bin/bash
# round 3, GPU session 28: step budget of one rank of an N-GPU run of BASELINE config 4 (the named 8-GPU configuration): gather schedule's multiply, colpipe compute path
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s28; mkdir -p $O
( timeout 900 python tools/rank_step_budget.py --workload config4_131072_23x23_fill1_fp64 --ranks 1,2,4,8 --steps 5
  timeout 900 python tools/rank_step_budget.py --workload config4_131072_23x23_fill1_fp64 --ranks 2,4,8 --steps 5 --colpipe 8 | sed "s/^# workload/# colpipe, 8 column chunks, two compute streams; workload/" ) 2>&1 | grep -v amdgpu.ids | tee $O/rank_step_budget_config4.txt
