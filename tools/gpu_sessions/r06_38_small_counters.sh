#!/bin/bash
# round 6, GPU session 38: parity of the small-block kernel again, then its counters (sizes 5 and 8, four products in flight) next to the packed 4 x 4 kernel's
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s38; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_small_blocks.py -q -m gpu -x 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest.txt
DBCSR_AMD_SWEEP_PLAIN=600 DBCSR_AMD_SWEEP_FORCED=200 timeout 900 python -m pytest tests/test_gpu_random_sweep.py -q -m gpu -x -n 4 -k "matches_oracle or forced" 2>&1 | grep -v "$F" | tail -3 | tee -a $O/pytest.txt
B=$(python3 -c 'import json; print(json.dumps([{"mix": "1,%d" % s, "fill": 0.1, "size": 1425 * s} for s in (4, 5, 8)]))')
export DBCSR_AMD_MM_SMALL=4
bash tools/profile_cmd.sh r06_small_blocks python $PWD/tools/block_bench.py --label counters --batch "$B" > $O/prof.log 2>&1
tail -60 gpurun_out/prof_r06_small_blocks/summary.txt 2>/dev/null | cut -c1-200
cp gpurun_out/prof_r06_small_blocks/summary.txt $O/prof_summary.txt 2>/dev/null
