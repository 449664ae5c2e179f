#!/bin/bash
# round 5, GPU session 10: config 5 at full size with the counter passes of bench.py (the default line leaves them out: two passes of 131072^2 break its budget)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s10; mkdir -p $O
( time timeout 900 python bench.py --workload config5_131072_32x32_fill20_fp32 --steps 2 --warmup 1 --cpu-seconds 0 --no-other-configs ) > $O/bench_config5.json 2> $O/bench_config5.err
tail -3 $O/bench_config5.err; cut -c1-1800 $O/bench_config5.json
