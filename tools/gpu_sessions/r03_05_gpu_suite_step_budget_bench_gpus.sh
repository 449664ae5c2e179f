#!/bin/bash
# round 3, GPU session 5: whole GPU suite, per-rank step budget, bench --gpus 2/4 on the one GPU (gloo)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s05; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1
tail -12 $O/pytest_gpu.txt
timeout 600 python tools/rank_step_budget.py > $O/rank_step_budget_plan1.txt 2>&1; grep -v amdgpu.ids $O/rank_step_budget_plan1.txt
DBCSR_AMD_MM_PLAN=0 timeout 600 python tools/rank_step_budget.py > $O/rank_step_budget_plan0.txt 2>&1; grep -v amdgpu.ids $O/rank_step_budget_plan0.txt
for n in 2 4; do
  timeout 900 python bench.py --gpus $n --steps 3 --warmup 1 --no-pmc --cpu-seconds 0 > $O/bench_gpus$n.json 2> $O/bench_gpus$n.err
  echo "gpus $n rc $?"; tail -c 900 $O/bench_gpus$n.json; echo
done
