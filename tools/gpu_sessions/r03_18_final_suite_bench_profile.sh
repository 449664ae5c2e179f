#!/bin/bash
# round 3, GPU session 18: the whole GPU suite, smoke(), the default bench.py line, rocprofv3 kernel trace + counter passes of the same command
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s18; mkdir -p $O
SECONDS=0
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $? after ${SECONDS}s"; tail -4 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
SECONDS=0
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? wall ${SECONDS}s"; tail -c 3000 $O/bench_default.json
bash tools/profile_bench.sh r03_final > $O/profile.log 2>&1; tail -5 $O/profile.log; ls gpurun_out/prof_r03_final | head
