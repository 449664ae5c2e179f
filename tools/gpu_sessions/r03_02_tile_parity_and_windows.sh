#!/bin/bash
# round 3, GPU session 2: tile kernel parity, then config 2 with the tile kernel at several team windows
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile_kernel.py -x -q -m gpu > $O/pytest_tile.txt 2>&1
tail -25 $O/pytest_tile.txt
if grep -q "failed\|error" $O/pytest_tile.txt; then echo "TILE TESTS FAILED: skipping the timing"; fi
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1], round(d["ms_per_step"],3), "ms/step; kernel", round(r["kernel_ms"],3), r["kernel"], "frac", round(r["frac"],4), "fill", round(r["fill_products_ms"],3), "parity", d.get("parity_max_rel_err_vs_cpu_sample"), "traffic", r.get("traffic"), "hit", r.get("l2_hit_rate"), "mfma", r.get("mfma_busy_frac"), "sclk", r.get("sclk_mhz"))
except Exception as e:
    print(sys.argv[1], "unreadable", e, open(sys.argv[1]).read()[-600:])
PY
}
for w in 256 0 64 128 512 1024 4096; do
  DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_WINDOW=$w timeout 300 python bench.py --steps 5 --warmup 1 --no-pmc --cpu-seconds 2 > $O/bench_tile_w$w.json 2> $O/bench_tile_w$w.err
  show $O/bench_tile_w$w.json
done
DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_WINDOW=256 DBCSR_AMD_MM_TILE_RDV=1 timeout 300 python bench.py --steps 5 --warmup 1 --no-pmc --cpu-seconds 0 > $O/bench_tile_rdv.json 2> $O/bench_tile_rdv.err
show $O/bench_tile_rdv.json
DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_WINDOW=256 timeout 600 python bench.py --steps 5 --warmup 1 --pmc --cpu-seconds 0 > $O/bench_tile_pmc.json 2> $O/bench_tile_pmc.err
show $O/bench_tile_pmc.json
tail -3 $O/bench_tile_pmc.err
