#!/bin/bash
# waves per workgroup (DBCSR_AMD_MM_WG_WAVES = 1 / 2 / 4) on configs 2, 3, 4
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-pmc $BA > $O/bench_$name.json 2> $O/bench_$name.err; python - $O/bench_$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d['ms_per_step'], 3), round(d['value']), round(d['roofline']['kernel_ms'], 3), d['roofline']['kernel'][:40])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for wl in config3_32768_mixed13_23_32_fill5_fp64 config2_32768_23x23_fill10_fp64 config4_131072_23x23_fill1_fp64; do
  BA="--workload $wl"
  for w in 4 2 1; do run ${wl%%_*}_wgw$w DBCSR_AMD_MM_WG_WAVES=$w; done
done
