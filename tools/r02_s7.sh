#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 5 --warmup 1 --cpu-seconds 2 --no-pmc $BA > $O/bench_$name.json 2> $O/bench_$name.err; python - $O/bench_$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d['ms_per_step'], 3), round(d['value']), round(d['roofline']['kernel_ms'], 3), d['roofline']['kernel'][:40], d.get('parity_max_rel_err_vs_cpu_sample'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for ap in 0 1; do for sb in 0 1; do
  BA="--workload config3_32768_mixed13_23_32_fill5_fp64"
  run c3_ap${ap}_sb${sb} DBCSR_AMD_JIT_DEFS="-DDBCSR_EXACT_ALL_PIECES=$ap -DDBCSR_EXACT_SCHED_BARRIER=$sb"
  BA="--workload config2_32768_23x23_fill10_fp64"
  run c2_ap${ap}_sb${sb} DBCSR_AMD_MM_CLASSES=2 DBCSR_AMD_JIT_DEFS="-DDBCSR_EXACT_ALL_PIECES=$ap -DDBCSR_EXACT_SCHED_BARRIER=$sb"
done; done
