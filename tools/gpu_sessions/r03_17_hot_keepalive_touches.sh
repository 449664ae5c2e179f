#!/bin/bash
# round 3, GPU session 17: exact-size kernel whose products also touch the neighbouring A blocks of the block row (variants 3 / 4): fabric bytes and time
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s17; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernel_variants.py -q -m gpu -x -k "HOT_VARIANT" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in 0 3 4 0 3 4; do
  DBCSR_AMD_MM_HOT_VARIANT=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-pmc --cpu-seconds 0 > $O/b_v$v.json 2> $O/b_v$v.err
  python - $O/b_v$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("variant %s: %.3f ms/step  kernel %.3f ms  frac %.4f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["frac"]))
PY
done
for v in 0 3 4; do
  DBCSR_AMD_MM_HOT_VARIANT=$v timeout 900 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > $O/pmc_v$v.json 2> $O/pmc_v$v.err
  python - $O/pmc_v$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("variant %s with counters: kernel %.3f ms  traffic %.1f GB  l2 hit %s  mfma busy %s  sclk %s" % (sys.argv[2], r["kernel_ms"], (r["traffic"] or 0)/1e9, r.get("l2_hit_rate"), r.get("mfma_busy_frac"), r.get("sclk_mhz")))
PY
done
