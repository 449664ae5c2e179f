#!/bin/bash
# A/B on ONE box: the library of commit a23725e (before work records / waves per workgroup) against the current one, config 2
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/s20; mkdir -p $O
run() { dir=$1; name=$2; shift 2; ( cd $dir && env "$@" timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-pmc --workload config2_32768_23x23_fill10_fp64 > $O/bench_$name.json 2> $O/bench_$name.err ); python - $O/bench_$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d['ms_per_step'], 3), round(d['value']), round(d['roofline']['kernel_ms'], 3), d['roofline']['kernel'][:40])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run $R/_ab/old old_1
run $R new_default_1
run $R new_w0_wg4 DBCSR_AMD_MM_WORK=0 DBCSR_AMD_MM_WG_WAVES=4
run $R/_ab/old old_2
run $R new_default_2
