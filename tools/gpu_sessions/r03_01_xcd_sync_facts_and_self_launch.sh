#!/bin/bash
# round 3, GPU session 1: XCD sync facts, exact-size kernel with unpaired fragment reads, bench.py --gpus 2 from a bare shell
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s01; mkdir -p $O
( cd tools/ubench && timeout 120 ./ubench_xcd_sync ) > $O/ubench_xcd_sync.txt 2>&1
for rep in 1 2; do
  for v in 0 2; do
    DBCSR_AMD_MM_HOT_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-pmc --cpu-seconds 0 > $O/bench_var${v}_rep$rep.json 2> $O/bench_var${v}_rep$rep.err
  done
done
DBCSR_AMD_MM_DBG=1 timeout 300 python bench.py --steps 5 --warmup 1 --no-pmc --cpu-seconds 0 > $O/bench_dbg1.json 2>&1
timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --workload mid_8192_23x23_fill10_fp64 --cpu-seconds 2 > $O/bench_gpus2.json 2> $O/bench_gpus2.err
echo "gpus2 rc $?" >> $O/bench_gpus2.err
timeout 600 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_multiply.py -x -q -m gpu > $O/pytest_subset.txt 2>&1
tail -3 $O/pytest_subset.txt
cat $O/ubench_xcd_sync.txt
for f in $O/bench_var*.json $O/bench_dbg1.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"],3), "ms/step; kernel", round(d["roofline"]["kernel_ms"],3), d["roofline"]["kernel"], round(d["roofline"]["frac"],4))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -c 1500 $O/bench_gpus2.json; tail -5 $O/bench_gpus2.err
