#!/bin/bash
# round 3, GPU session 37: counters of the persistent form of the exact-size kernel (why is it slower?)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s37; mkdir -p $O
DBCSR_AMD_MM_HOT_PERSISTENT=1 timeout 900 python bench.py --steps 6 --warmup 2 --cpu-seconds 0 > $O/bench_persistent_pmc.json 2> $O/err.txt
python - $O/bench_persistent_pmc.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print({k: r.get(k) for k in ("kernel","kernel_ms","traffic","l2_hit_rate","mfma_busy_frac","sclk_mhz","fabric_tb_per_s")})
PY
