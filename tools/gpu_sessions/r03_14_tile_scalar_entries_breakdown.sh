#!/bin/bash
# round 3, GPU session 14: tile kernel after the scalar entry loads / uniform epilogue: parity, time breakdown, timing
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile_kernel.py tests/test_gpu_plan_reuse.py -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
cat > /tmp/tile_stats.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from dbcsr_amd import randmat
from dbcsr_amd.multiply import MultiplyEngine
eng = MultiplyEngine()
A, B, C = randmat.perf_matrices(32768, 32768, 32768, (0.9, 0.9, 0.9), [1, 23], [1, 23], [1, 23], dtype=torch.float64, engine=eng)
for _ in range(3):
    out, counts = eng.multiply_local(1.0, A, B, 1.0, C)
torch.cuda.synchronize()
print("window", os.environ.get("DBCSR_AMD_MM_TILE_WINDOW"), "knobs", os.environ.get("DBCSR_AMD_MM_TILE_KNOBS"), "kernel ms %.3f" % eng.last_timing()[1], eng.last_kernel(), eng.tile_stats())
PY
for w in 0 128 256 512; do
  DBCSR_AMD_MM_TILE_VERBOSE=1 DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_KNOBS=32 DBCSR_AMD_MM_TILE_WINDOW=$w timeout 300 python /tmp/tile_stats.py 2>&1 | grep -v amdgpu.ids | tail -3
done | tee $O/tile_time_breakdown.txt
for w in 128 192 256 384 512; do
  DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_KNOBS=0 DBCSR_AMD_MM_TILE_WINDOW=$w timeout 300 python /tmp/tile_stats.py 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $O/tile_windows.txt
