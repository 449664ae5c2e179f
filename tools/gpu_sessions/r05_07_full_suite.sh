#!/bin/bash
# round 5, GPU session 7: the whole -m gpu suite, smoke(), the kernel trace + stats and the counter passes of the bench command, the default bench line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s07; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -30 ) > $O/pytest_gpu.txt 2>&1
tail -8 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -2 > $O/smoke.txt; cat $O/smoke.txt
PROFILE_PASSES=4 timeout 420 bash tools/profile_bench.sh r05_final > $O/profile_summary.txt 2>&1
cp gpurun_out/prof_r05_final/summary.txt $O/summary.txt 2>/dev/null
find gpurun_out/prof_r05_final/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -6 $O/kernel_stats.csv | cut -c1-260
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err; cut -c1-700 $O/bench_default.json
