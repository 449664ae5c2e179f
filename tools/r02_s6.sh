#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s6; mkdir -p $O
export DBCSR_AMD_MM_VERBOSE=1
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_comm.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt
unset DBCSR_AMD_MM_VERBOSE
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-pmc $BA > $O/bench_$name.json 2> $O/bench_$name.err; }
BA="--workload config3_32768_mixed13_23_32_fill5_fp64"
run c3_classes DBCSR_AMD_MM_CLASSES=1
run c3_pipe DBCSR_AMD_MM_CLASSES=0
BA="--workload config2_32768_23x23_fill10_fp64"
run c2_hot DBCSR_AMD_MM_CLASSES=1
run c2_class DBCSR_AMD_MM_CLASSES=2
BA="--workload mid_16384_23x23_fill10_fp64"
run mid_hot DBCSR_AMD_MM_CLASSES=1
for f in c3_classes c3_pipe c2_hot c2_class mid_hot; do python - $O/bench_$f.json $f <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d['ms_per_step'], 3), round(d['value']), round(d['roofline']['kernel_ms'], 3), d['roofline']['kernel'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
tail -3 $O/bench_c3_classes.err
