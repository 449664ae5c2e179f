#!/bin/bash
# round 3, GPU session 24: plan reuse by address (dbcsr_amd_mm_trust_plan): parity, then the colpipe compute path of one rank again
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plan_reuse.py tests/test_gpu_cannon_shared_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
( for nc in 2 4 8; do timeout 900 python tools/rank_step_budget.py --ranks 4,8 --colpipe $nc | sed "s/^# workload/# colpipe, $nc column chunks, plans reused by address; workload/"; done ) 2>&1 | grep -v amdgpu.ids | tee $O/rank_step_budget_colpipe_trusted.txt
