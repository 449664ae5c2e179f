#!/bin/bash
# round 3, GPU session 8: resident Fortran loop per-multiply times; tile kernel protocol knobs
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s08; mkdir -p $O
( cd /tmp && OMP_NUM_THREADS=8 MKL_THREADING_LAYER=SEQUENTIAL timeout 900 $GRAFT_REPO_ROOT/oracle/_ref/host_resident/dbcsr_resident_loop 32768 0.9 23 10 0 2>&1 | grep "resident_loop" ) > $O/resident_loop_config2.txt 2>&1
cat $O/resident_loop_config2.txt
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
for kn in 0 5 8 13 16 29; do for w in 256 512; do
  DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_KNOBS=$kn DBCSR_AMD_MM_TILE_WINDOW=$w timeout 300 python bench.py --steps 5 --warmup 1 --no-pmc --cpu-seconds 0 > $O/bench_tile_k${kn}_w$w.json 2> $O/bench_tile_k${kn}_w$w.err
  show $O/bench_tile_k${kn}_w$w.json
done; done
