#!/bin/bash
# round 3, GPU session 11: panel size x waves per workgroup of the exact-size kernel on config 2, two repetitions each
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s11; mkdir -p $O
for rep in 1 2; do for mb in 160 200 256 320 400; do for ww in 1 2 4; do
  DBCSR_AMD_MM_PANEL_MB=$mb DBCSR_AMD_MM_WG_WAVES=$ww timeout 300 python bench.py --steps 8 --warmup 2 --no-pmc --cpu-seconds 0 > $O/b_${mb}_${ww}_$rep.json 2> $O/b_${mb}_${ww}_$rep.err
  python - $O/b_${mb}_${ww}_$rep.json $mb $ww $rep <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("panel %s MB  waves/wg %s  rep %s: %.3f ms/step  kernel %.3f ms  frac %.4f" % (sys.argv[2], sys.argv[3], sys.argv[4], d["ms_per_step"], r["kernel_ms"], r["frac"]))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done; done; done | tee $O/sweep.txt
