#!/bin/bash
# class-kernel tuning switches again, now with one wave per workgroup (config 3); bench lines with PMC for configs 3 and 2
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s26; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-pmc $BA > $O/bench_$name.json 2> $O/bench_$name.err; python - $O/bench_$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d['ms_per_step'], 3), round(d['value']), round(d['roofline']['kernel_ms'], 3), d['roofline']['kernel'][:40])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
BA="--workload config3_32768_mixed13_23_32_fill5_fp64"
run c3_default
run c3_ap0_sb1 "DBCSR_AMD_JIT_DEFS=-DDBCSR_EXACT_ALL_PIECES=0"
run c3_ap1_sb0 "DBCSR_AMD_JIT_DEFS=-DDBCSR_EXACT_SCHED_BARRIER=0"
run c3_ap0_sb0 "DBCSR_AMD_JIT_DEFS=-DDBCSR_EXACT_ALL_PIECES=0 -DDBCSR_EXACT_SCHED_BARRIER=0"
run c3_panel80 DBCSR_AMD_MM_PANEL_MB=80
run c3_panel240 DBCSR_AMD_MM_PANEL_MB=240
timeout 900 python bench.py --workload config3_32768_mixed13_23_32_fill5_fp64 --cpu-seconds 0 > $O/bench_c3_pmc.json 2> $O/bench_c3_pmc.err
python - $O/bench_c3_pmc.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(json.dumps(d["roofline"]))
PY
