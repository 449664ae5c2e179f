#!/bin/bash
# round 3, GPU session 26: colpipe with the chunks alternating between two compute streams: parity (ranks sharing the GPU), compute path of one rank
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s26; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cannon_shared_gpu.py -q -m gpu -x -k colpipe > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
( for st in 1 2; do for nc in 4 8; do DBCSR_AMD_COLPIPE_STREAMS=$st timeout 900 python tools/rank_step_budget.py --ranks 4,8 --colpipe $nc | sed "s/^# workload/# colpipe, $nc column chunks, $st compute stream(s); workload/"; done; done ) 2>&1 | grep -v amdgpu.ids | tee $O/rank_step_budget_colpipe_streams.txt
