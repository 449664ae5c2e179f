#!/bin/bash
# round 3, GPU session 29: colpipe compute path without record_stream on the output buffer: configs 2 and 4
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s29; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cannon_shared_gpu.py -q -m gpu -x -k colpipe > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
( for wl in config2_32768_23x23_fill10_fp64 config4_131072_23x23_fill1_fp64; do for st in 2 1; do
  DBCSR_AMD_COLPIPE_STREAMS=$st timeout 900 python tools/rank_step_budget.py --workload $wl --ranks 2,4,8 --steps 6 --colpipe 8 | sed "s/^# workload/# colpipe, 8 column chunks, $st compute stream(s); workload/"; done; done ) 2>&1 | grep -v amdgpu.ids | tee $O/rank_step_budget_colpipe_final.txt
