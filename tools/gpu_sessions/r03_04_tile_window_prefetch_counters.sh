#!/bin/bash
# round 3, GPU session 4: tile kernel -- window, L2 prefetch, wait statistics, counters
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile_kernel.py -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["ms_per_step"],3), "ms/step; kernel", round(r["kernel_ms"],3), r["kernel"][:24], "frac", round(r["frac"],4), "parity", d.get("parity_max_rel_err_vs_cpu_sample"), "traffic", r.get("traffic"), "hit", r.get("l2_hit_rate"), "mfma", r.get("mfma_busy_frac"), "sclk", r.get("sclk_mhz"))
except Exception as e:
    print(sys.argv[1], "unreadable", e, open(sys.argv[1]).read()[-600:])
PY
}
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
print("kernel ms", eng.last_timing()[1], eng.last_kernel(), eng.tile_stats())
PY
for pf in 0 1; do for w in 192 256 384; do
  DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_PREFETCH=$pf DBCSR_AMD_MM_TILE_WINDOW=$w timeout 300 python bench.py --steps 5 --warmup 1 --no-pmc --cpu-seconds 0 > $O/bench_tile_pf${pf}_w$w.json 2> $O/bench_tile_pf${pf}_w$w.err
  show $O/bench_tile_pf${pf}_w$w.json
done; done
for pf in 0 1; do
  DBCSR_AMD_MM_TILE_VERBOSE=1 DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_PREFETCH=$pf DBCSR_AMD_MM_TILE_WINDOW=256 timeout 300 python /tmp/tile_stats.py 2>&1 | grep -v amdgpu.ids | tail -3
done
DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_PREFETCH=0 DBCSR_AMD_MM_TILE_WINDOW=256 timeout 600 python bench.py --steps 5 --warmup 1 --pmc --cpu-seconds 0 > $O/bench_tile_pf0_pmc.json 2> $O/bench_tile_pf0_pmc.err; show $O/bench_tile_pf0_pmc.json
DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_PREFETCH=1 DBCSR_AMD_MM_TILE_WINDOW=256 timeout 600 python bench.py --steps 5 --warmup 1 --pmc --cpu-seconds 0 > $O/bench_tile_pf1_pmc.json 2> $O/bench_tile_pf1_pmc.err; show $O/bench_tile_pf1_pmc.json
