#!/bin/bash
# round 6, GPU session 18: the default bench line (full contract: roofline + cpu_baseline + other configurations with counters) and the rocprofv3
# kernel-trace / counter summaries of the same command
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s18; mkdir -p $O
( time timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err; cut -c1-600 $O/bench_default.json
PROFILE_PASSES=4 timeout 900 bash tools/profile_bench.sh r06_final > $O/profile_summary.txt 2>&1
tail -30 $O/profile_summary.txt | cut -c1-250
cp gpurun_out/prof_r06_final/summary.txt $O/prof_summary.txt 2>/dev/null
find gpurun_out/prof_r06_final/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; 2>/dev/null
