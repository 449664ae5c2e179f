#!/bin/bash
# round 4, GPU session 17: the whole -m gpu suite, smoke() and the default bench line on the final tree
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s17; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -15 > $O/full_gpu_suite.txt; cat $O/full_gpu_suite.txt
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -6 $O/smoke.txt
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/bench_default.err; cut -c1-400 $O/bench_default.json
