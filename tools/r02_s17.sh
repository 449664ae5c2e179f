#!/bin/bash
# waves per workgroup for the fp32 kernels and the generic fp64 LDS kernel
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_multiply.py -x -q -m gpu > $O/pytest.txt 2>&1
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
for wl in fp32_16384_32x32_fill20 config5_131072_32x32_fill20_fp32; do
  BA="--workload $wl"
  for w in 4 1; do run ${wl%%_*}_wgw$w DBCSR_AMD_MM_WG_WAVES=$w; done
done
BA="--workload config3_32768_mixed13_23_32_fill5_fp64"
for w in 4 1; do run c3_generic_wgw$w DBCSR_AMD_MM_WG_WAVES=$w DBCSR_AMD_MM_CLASSES=0 DBCSR_AMD_MM_KERNEL=lds1; done
