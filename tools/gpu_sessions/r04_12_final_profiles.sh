#!/bin/bash
# round 4, GPU session 12: the profiles of the final state -- rocprofv3 kernel trace + stats and the counter passes of the bench command, the default bench line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s12; mkdir -p $O
bash tools/profile_bench.sh r04_final > $O/profile_summary.txt 2>&1
cp gpurun_out/prof_r04_final/summary.txt $O/summary.txt 2>/dev/null
find gpurun_out/prof_r04_final/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
head -12 $O/kernel_stats.csv
