#!/bin/bash
# round 4, GPU session 13: config 3 (and 4), where the time of the class launches goes: rocprofv3 kernel trace per class kernel, registers / LDS of each
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s13; mkdir -p $O
export TMPDIR=/tmp
for w in config3_32768_mixed13_23_32_fill5_fp64 config4_131072_23x23_fill1_fp64; do
( cd /tmp && DBCSR_AMD_MM_VERBOSE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/trace_$w -o t --output-format csv -- python $OLDPWD/bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 0 --no-pmc --no-other-configs ) > $O/$w.log 2>&1
grep "compiled class kernel" $O/$w.log | sort | uniq | head -12
python3 - $O/trace_$w <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "mm_numeric" in n or "class" in n:
            agg[(n[:70], r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-70s vgpr %s agpr %s lds %s grid %s wg %s  calls %d  mean %.3f ms  min %.3f" % (k + (len(v), sum(v) / len(v), min(v))))
PY
done 2>&1 | tee $O/summary.txt
find $O -name "*.csv" -size +1M -delete
