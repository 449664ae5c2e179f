#!/bin/bash
# round 3, GPU session 16: C write pattern by itself (config 4), class launches of config 3 on several streams
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s16; mkdir -p $O
timeout 300 tools/ubench/ubench_c_write > $O/ubench_c_write.txt 2>&1; cat $O/ubench_c_write.txt
timeout 600 python -m pytest tests/test_gpu_kernel_variants.py -q -m gpu -x -k "CLASS_STREAMS or fp64_variant_matches" > $O/pytest_classes.txt 2>&1; tail -3 $O/pytest_classes.txt
for ns in 1 2 3 4; do
  wl=config3_32768_mixed13_23_32_fill5_fp64
  DBCSR_AMD_MM_CLASS_STREAMS=$ns timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-pmc --cpu-seconds 0 > $O/b_config3_s$ns.json 2> $O/b_config3_s$ns.err
  python - $O/b_config3_s$ns.json $ns <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("config3 class streams %s: %.3f ms/step  kernel %.3f ms  %.1f GFLOP/s  parity %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], d["value"], d.get("parity_max_rel_err_vs_cpu_sample")))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
