#!/bin/bash
# is it the box?  fabric micro-benchmark + config 2 in the morning's configuration (WORK=0, 4 waves per workgroup) + clocks; config 3 PMC
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s19; mkdir -p $O
rocm-smi --showclocks --showpower --showperflevel > $O/smi_before.txt 2>&1
timeout 300 ./tools/ubench/ubench_fabric windows > $O/ubench_fabric_windows.txt 2>&1
grep -E "reg\+ds_write A\+B \(16" $O/ubench_fabric_windows.txt
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-pmc $BA > $O/bench_$name.json 2> $O/bench_$name.err; python - $O/bench_$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d['ms_per_step'], 3), round(d['value']), round(d['roofline']['kernel_ms'], 3), d['roofline']['kernel'][:40])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
BA="--workload config2_32768_23x23_fill10_fp64"
run c2_old DBCSR_AMD_MM_WORK=0 DBCSR_AMD_MM_WG_WAVES=4
run c2_new
rocm-smi --showclocks --showpower > $O/smi_after.txt 2>&1
grep -E "sclk|mclk|fclk|socclk|Power" $O/smi_after.txt | head
timeout 900 python bench.py --workload config3_32768_mixed13_23_32_fill5_fp64 --cpu-seconds 0 > $O/bench_c3_pmc.json 2> $O/bench_c3_pmc.err
tail -c 900 $O/bench_c3_pmc.json
