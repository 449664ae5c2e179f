#!/bin/bash
# round 3, GPU session 35: the other BASELINE configurations on one GPU with the final library (bench lines without counter passes / CPU baseline)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s35; mkdir -p $O
for wl in config1_4096_4x4_fill10_fp64 config3_32768_mixed13_23_32_fill5_fp64 config4_131072_23x23_fill1_fp64 config5_131072_32x32_fill20_fp32; do
  SECONDS=0
  timeout 900 python bench.py --workload $wl --steps 5 --warmup 2 --no-pmc --cpu-seconds 0 > $O/b_$wl.json 2> $O/b_$wl.err
  python - $O/b_$wl.json $wl $SECONDS <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("%s: %.3f ms/step  kernel %.3f ms  %.1f GFLOP/s  bound %s frac %.3f  k_passes %s  %s  (wall %ss)" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], d["value"], r["bound"], r["frac"], d["config"].get("k_passes"), r["kernel"][:60], sys.argv[3]))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done | tee $O/other_configs.txt
