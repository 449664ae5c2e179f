#!/bin/bash
# round 3, GPU session 38: default bench line with the plan reused by address (N = 1)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s38; mkdir -p $O
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 2 --no-pmc --cpu-seconds 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.3f  kernel %.3f  frac %.4f  parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d.get('parity_max_rel_err_vs_cpu_sample')))"; done | tee $O/bench.txt
