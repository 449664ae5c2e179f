#!/bin/bash
# round 4, GPU session 14: the 4 x 4 kernel (config 1) with 16 entries per request and 8 / 16 products' elements in flight per wave:
# parity (variants, plan reuse, random sweep's tiny cases), then config 1 at full size, both depths
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s14; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_plan_reuse.py -q -k "TINY or tiny" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $O/tests.txt; cat $O/tests.txt
W=config1_4096_4x4_fill10_fp64
for t in 1 2; do
  DBCSR_AMD_MM_TINY=$t timeout 300 python bench.py --workload $W --steps 50 --warmup 5 --cpu-seconds 0 --no-pmc --no-other-configs > $O/bench_tiny$t.json 2> $O/bench_tiny$t.err
  python3 - $O/bench_tiny$t.json $t <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("TINY=%s  ms_per_step %.4f  kernel_ms %.4f  fill_products_ms %s  value %.1f %s  frac %.3f (%s)  parity %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r.get("fill_products_ms"), d["value"], d["unit"], r["frac"], r["bound"], d.get("parity_max_rel_err_vs_cpu_sample")))
PY
done 2>&1 | tee $O/summary.txt
