#!/bin/bash
# round 4, GPU session 22: the clean rebuild of the final tree: smoke(), the kernel-variant / plan-reuse / multiply tests, config 1 and config 2 steps
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s22; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_plan_reuse.py tests/test_gpu_multiply.py -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $O/tests.txt
for w in config1_4096_4x4_fill10_fp64 config2_32768_23x23_fill10_fp64; do
  timeout 200 python bench.py --workload $w --steps 10 --warmup 2 --cpu-seconds 0 --no-pmc --no-other-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'], round(d['ms_per_step'],4), 'ms', round(d['value'],1), d['unit'], 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['roofline']['kernel'])"
done | tee $O/bench.txt
