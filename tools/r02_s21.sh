#!/bin/bash
# A/B on ONE box after the scratch fix: library of commit a23725e against the current one.
# The old tree is made first (it is not kept in the repository):
#   mkdir -p _ab/old && git archive a23725e dbcsr_amd bench.py include oracle/__init__.py oracle/oracle.py | tar -x -C _ab/old && make -C _ab/old/dbcsr_amd/csrc
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/s21; mkdir -p $O
run() { dir=$1; name=$2; wl=$3; shift 3; ( cd $dir && env "$@" timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-pmc --workload $wl > $O/bench_$name.json 2> $O/bench_$name.err ); python - $O/bench_$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d['ms_per_step'], 3), round(d['value']), round(d['roofline']['kernel_ms'], 3), d['roofline']['kernel'][:40])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_multiply.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -2 $O/pytest.txt
C2=config2_32768_23x23_fill10_fp64; C3=config3_32768_mixed13_23_32_fill5_fp64; C4=config4_131072_23x23_fill1_fp64
run $R/_ab/old c2_old $C2
run $R c2_new $C2
run $R c2_new_wg4 $C2 DBCSR_AMD_MM_WG_WAVES=4
run $R c2_new_wg2 $C2 DBCSR_AMD_MM_WG_WAVES=2
run $R/_ab/old c3_old $C3
run $R c3_new $C3
run $R/_ab/old c4_old $C4
run $R c4_new $C4
run $R c4_new_wg4 $C4 DBCSR_AMD_MM_WG_WAVES=4
