#!/bin/bash
# round 5, GPU session 16: bench.py of the final tree at 2 ranks on the one device (the provisional line on stderr) and at its defaults
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s16; mkdir -p $O
( time timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 --cpu-seconds 0 --no-pmc ) > $O/bench_gpus2.json 2> $O/bench_gpus2.err
grep -c BENCH_PROVISIONAL $O/bench_gpus2.err; grep '^{"metric"' $O/bench_gpus2.json | cut -c1-400
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err; cut -c1-400 $O/bench_default.json
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
