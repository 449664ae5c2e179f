#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s4; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -40 $O/pytest_gpu.txt
for mb in 160 64 32; do
  env DBCSR_AMD_MM_PANEL_MB=$mb timeout 400 python bench.py --workload config4_131072_23x23_fill1_fp64 --steps 3 --warmup 1 --cpu-seconds 0 > $O/bench_c4_panel$mb.json 2> $O/bench_c4_panel$mb.err
done
cat $O/bench_c4_*.json | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])
"
