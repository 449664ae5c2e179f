#!/bin/bash
# round 3, GPU session 3: tile kernel with rationed progress stores; plan reuse
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile_kernel.py tests/test_gpu_plan_reuse.py -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["ms_per_step"],3), "ms/step; kernel", round(r["kernel_ms"],3), r["kernel"][:24], "frac", round(r["frac"],4), "fill", round(r["fill_products_ms"],3), "parity", d.get("parity_max_rel_err_vs_cpu_sample"), "traffic", r.get("traffic"), "hit", r.get("l2_hit_rate"), "mfma", r.get("mfma_busy_frac"), "sclk", r.get("sclk_mhz"))
except Exception as e:
    print(sys.argv[1], "unreadable", e, open(sys.argv[1]).read()[-600:])
PY
}
# the default path with plan reuse (whole-multiply time is what changes)
timeout 300 python bench.py --steps 10 --warmup 2 --no-pmc --cpu-seconds 2 > $O/bench_hot_plan.json 2> $O/bench_hot_plan.err; show $O/bench_hot_plan.json
DBCSR_AMD_MM_PLAN=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-pmc --cpu-seconds 0 > $O/bench_hot_noplan.json 2> $O/bench_hot_noplan.err; show $O/bench_hot_noplan.json
for pub in 0 1; do for w in 128 256 512 1024; do
  DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_PUB=$pub DBCSR_AMD_MM_TILE_WINDOW=$w timeout 300 python bench.py --steps 5 --warmup 1 --no-pmc --cpu-seconds 0 > $O/bench_tile_p${pub}_w$w.json 2> $O/bench_tile_p${pub}_w$w.err
  show $O/bench_tile_p${pub}_w$w.json
done; done
DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_WINDOW=0 timeout 600 python bench.py --steps 5 --warmup 1 --pmc --cpu-seconds 0 > $O/bench_tile_w0_pmc.json 2> $O/bench_tile_w0_pmc.err; show $O/bench_tile_w0_pmc.json
DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_WINDOW=256 timeout 600 python bench.py --steps 5 --warmup 1 --pmc --cpu-seconds 2 > $O/bench_tile_w256_pmc.json 2> $O/bench_tile_w256_pmc.err; show $O/bench_tile_w256_pmc.json
