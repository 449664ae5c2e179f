#!/bin/bash
# round 6, GPU session 62: counters of the small-block kernel in its final form (two products in flight, resources prepared by the lanes), sizes 5 and 8
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s62; mkdir -p $O
B=$(python3 -c 'import json; print(json.dumps([{"mix": "1,%d" % s, "fill": 0.1, "size": 1425 * s} for s in (5, 8)]))')
bash tools/profile_cmd.sh r06_small_final python $PWD/tools/block_bench.py --label counters --batch "$B" > $O/prof.log 2>&1
grep -A12 "mm_numeric_f64_small" gpurun_out/prof_r06_small_final/summary.txt | cut -c1-160 | head -90
cp gpurun_out/prof_r06_small_final/summary.txt $O/prof_summary.txt
