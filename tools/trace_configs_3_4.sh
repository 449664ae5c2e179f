#!/bin/bash
# kernel traces of configs 4 and 3 (where does the time outside the numeric kernel go), occupancy counters of config 3
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; O=$R/gpurun_out/s22; mkdir -p $O
export TMPDIR=/tmp
for wl in config4_131072_23x23_fill1_fp64 config3_32768_mixed13_23_32_fill5_fp64; do
  t=${wl%%_*}
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$t -o trace --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-pmc --workload $wl ) > $O/trace_$t.log 2>&1
  f=$(find $O/trace_$t -name "*kernel_stats.csv" | head -1)
  cp $f $O/${t}_kernel_stats.csv
  python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-60s calls %5s  avg %10.1f us  total %9.2f ms  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_c3 -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-pmc --workload config3_32768_mixed13_23_32_fill5_fp64 ) > $O/pmc_c3.log 2>&1
python - $O/pmc_c3 <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in agg.items():
    if "class" in k or "hot" in k:
        n = max(1, cnt[k])
        print(k, "dispatches", n, {c: "%.3e" % (x / n) for c, x in v.items()})
        if v.get("GRBM_GUI_ACTIVE"):
            print("   waves per SIMD resident on average: %.2f   MFMA busy: %.3f" % (4 * v["SQ_WAVE_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024), v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
