#!/bin/bash
# round 3, GPU session 10: occupancy / panel sweeps of the exact-size kernel on config 2 (does a lower occupancy cut the A re-fetches?)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s10; mkdir -p $O
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["ms_per_step"],3), "ms/step; kernel", round(r["kernel_ms"],3), r["kernel"][:24], "frac", round(r["frac"],4), "traffic", r.get("traffic"), "hit", r.get("l2_hit_rate"), "mfma", r.get("mfma_busy_frac"), "sclk", r.get("sclk_mhz"))
except Exception as e:
    print(sys.argv[1], "unreadable", e, open(sys.argv[1]).read()[-600:])
PY
}
for pad in 0 1200 3400 6400; do
  DBCSR_AMD_MM_LDS_PAD=$pad timeout 300 python bench.py --steps 8 --warmup 2 --no-pmc --cpu-seconds 0 > $O/bench_pad$pad.json 2> $O/bench_pad$pad.err; show $O/bench_pad$pad.json
done
for ww in 2 4; do
  DBCSR_AMD_MM_WG_WAVES=$ww timeout 300 python bench.py --steps 8 --warmup 2 --no-pmc --cpu-seconds 0 > $O/bench_ww$ww.json 2> $O/bench_ww$ww.err; show $O/bench_ww$ww.json
done
for mb in 128 200 256; do
  DBCSR_AMD_MM_PANEL_MB=$mb timeout 300 python bench.py --steps 8 --warmup 2 --no-pmc --cpu-seconds 0 > $O/bench_panel$mb.json 2> $O/bench_panel$mb.err; show $O/bench_panel$mb.json
done
DBCSR_AMD_MM_LDS_PAD=3400 timeout 600 python bench.py --steps 5 --warmup 1 --pmc --cpu-seconds 0 > $O/bench_pad3400_pmc.json 2> $O/bench_pad3400_pmc.err; show $O/bench_pad3400_pmc.json
timeout 600 python bench.py --steps 5 --warmup 1 --pmc --cpu-seconds 0 > $O/bench_default_pmc.json 2> $O/bench_default_pmc.err; show $O/bench_default_pmc.json
