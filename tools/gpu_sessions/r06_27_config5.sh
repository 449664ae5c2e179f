#!/bin/bash
# round 6, GPU session 27: config 5 (fp32, 131072^2) again on another box -- the default bench line of session 18 had it at 1930 ms against 1836-1851 in round 5
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s27; mkdir -p $O
for i in 1 2; do
timeout 600 python -c "
import json, bench
r = bench.run_other_config('config5_131072_32x32_fill20_fp32', steps=3)
print('config5', r.get('ms_per_step'), r.get('value'), r.get('kernel_ms'), r.get('kernel'), r.get('k_passes'))" 2>/dev/null | grep config5 | tee -a $O/config5.txt
done
