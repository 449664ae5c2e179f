#!/bin/bash
# round 3, GPU session 22: where a wave of the shape-1 tile kernel spends its time
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s22; mkdir -p $O
for sh in 1 0; do for w in 0 512; do
  DBCSR_AMD_MM_TILE_VERBOSE=1 DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_SHAPE=$sh DBCSR_AMD_MM_TILE_KNOBS=32 DBCSR_AMD_MM_TILE_WINDOW=$w timeout 300 python tools/tile_stats.py 2>&1 | grep -v amdgpu.ids | tail -3
done; done | tee $O/breakdown.txt
