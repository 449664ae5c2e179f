#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py for the other BASELINE shapes (one GPU); leaves
# gpurun_out/prof_configs/<workload>_kernel_stats.csv (copy into profiles/).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; OUT=$R/gpurun_out/prof_configs; mkdir -p "$OUT"; export TMPDIR=/tmp
for w in config1_4096_4x4_fill10_fp64 config3_32768_mixed13_23_32_fill5_fp64 config4_131072_23x23_fill1_fp64 fp32_16384_32x32_fill20; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/$w" -o trace --output-format csv -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --cpu-seconds 0 ) > "$OUT/$w.log" 2>&1
  f=$(find "$OUT/$w" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -8 "$f" > "$OUT/${w}_kernel_stats.csv"
  grep -m1 "^{\"metric\"" "$OUT/$w.log" > "$OUT/${w}_bench.json"   # the bench line (rocprofv3 prints after it)
  rm -rf "$OUT/$w"
done
ls -la "$OUT"
