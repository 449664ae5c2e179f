#!/bin/bash
# round 3, GPU session 21: tile kernel shape 1 (4 x 3 C blocks per wave, one wave per SIMD, four-slot ring): parity, then config 2 by window
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile_kernel.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for sh in 1 0; do for w in 256 512 1024; do
  DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_SHAPE=$sh DBCSR_AMD_MM_TILE_WINDOW=$w timeout 600 python bench.py --steps 8 --warmup 2 --no-pmc --cpu-seconds 0 > $O/b_s${sh}_w$w.json 2> $O/b_s${sh}_w$w.err
  python - $O/b_s${sh}_w$w.json $sh $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("shape %s window %s: %.3f ms/step  kernel %.3f ms  frac %.4f  %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], r["kernel_ms"], r["frac"], r["kernel"]))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done; done
DBCSR_AMD_MM_TILE=2 DBCSR_AMD_MM_TILE_SHAPE=1 DBCSR_AMD_MM_TILE_WINDOW=512 DBCSR_AMD_MM_TILE_KNOBS=32 DBCSR_AMD_MM_TILE_VERBOSE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-pmc --cpu-seconds 0 2>&1 | grep "tile kernel" | tail -2
