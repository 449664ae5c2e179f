#!/bin/bash
# class launches on several streams (config 3); default bench line with PMC on this box
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-pmc $BA > $O/bench_$name.json 2> $O/bench_$name.err; python - $O/bench_$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d['ms_per_step'], 3), round(d['value']), round(d['roofline']['kernel_ms'], 3), d['roofline']['kernel'][:40])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
BA="--workload config3_32768_mixed13_23_32_fill5_fp64"
for n in 1 2 3 4; do run c3_streams$n DBCSR_AMD_MM_CLASS_STREAMS=$n; done
timeout 900 python bench.py --cpu-seconds 0 > $O/bench_default_pmc.json 2> $O/bench_default_pmc.err
tail -c 1200 $O/bench_default_pmc.json
