#!/bin/bash
# round 4, GPU session 15: the 4 x 4 kernel as a software pipeline over persistent waves (mm_numeric_f64_tiny_stream) against the
# wave-per-quad kernel, and B column panels sized for an XCD's L2 (config 1's B is 13 MB: one panel under the Infinity-Cache rule)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s15; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_plan_reuse.py -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $O/tests.txt; cat $O/tests.txt
W=config1_4096_4x4_fill10_fp64
for t in 1 2 3; do for p in 0 1 2 4; do
  if [ $p = 0 ]; then unset DBCSR_AMD_MM_PANEL_MB; else export DBCSR_AMD_MM_PANEL_MB=$p; fi
  DBCSR_AMD_MM_TINY=$t timeout 300 python bench.py --workload $W --steps 50 --warmup 5 --cpu-seconds 0 --no-pmc --no-other-configs > $O/bench_t${t}_p$p.json 2> $O/bench_t${t}_p$p.err
  python3 - $O/bench_t${t}_p$p.json $t $p <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("TINY=%s PANEL_MB=%s  ms_per_step %.4f  kernel_ms %.4f  %s  value %.1f %s  frac %.3f (%s)" % (sys.argv[2], sys.argv[3], d["ms_per_step"], r["kernel_ms"], r["kernel"], d["value"], d["unit"], r["frac"], r["bound"]))
except Exception as e:
    print("TINY=%s PANEL_MB=%s failed: %s" % (sys.argv[2], sys.argv[3], e))
PY
done; done 2>&1 | tee $O/summary.txt
