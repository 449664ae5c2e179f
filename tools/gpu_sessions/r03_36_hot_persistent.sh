#!/bin/bash
# round 3, GPU session 36: the exact-size kernel as persistent waves with a work counter per XCD: parity, config 2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s36; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernel_variants.py -q -m gpu -x -k "PERSISTENT" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in 0 1 0 1; do
  DBCSR_AMD_MM_HOT_PERSISTENT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('persistent $v: %.3f ms/step  kernel %.3f ms  frac %.4f  %s  parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel'], d.get('parity_max_rel_err_vs_cpu_sample')))"
done | tee $O/hot_persistent.txt
DBCSR_AMD_MM_HOT_PERSISTENT=1 DBCSR_AMD_MM_HOT_XCDS=0xee timeout 300 python bench.py --steps 6 --warmup 2 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('persistent, XCDs 0 and 4 idle (their C blocks not computed): kernel %.3f ms' % r['kernel_ms'])" | tee -a $O/hot_persistent.txt
