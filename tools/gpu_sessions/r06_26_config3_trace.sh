#!/bin/bash
# round 6, GPU session 26: config 3 launch by launch (rocprofv3 kernel trace: every class kernel / slab kernel with its registers, LDS, grid and time) and
# the counters of the same command (FETCH_SIZE, MFMA busy per kernel name), default build and DBCSR_AMD_MM_MID=0 beside it
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s26; mkdir -p $O
export TMPDIR=/tmp
w=config3_32768_mixed13_23_32_fill5_fp64
for M in 1 0; do
( cd /tmp && DBCSR_AMD_MM_MID=$M DBCSR_AMD_MM_VERBOSE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/trace_mid$M -o t --output-format csv -- python $OLDPWD/bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 0 --no-pmc --no-other-configs ) > $O/mid$M.log 2>&1
echo "== DBCSR_AMD_MM_MID=$M"
grep "compiled class kernel" $O/mid$M.log | sort | uniq | head -12
python3 - $O/trace_mid$M <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "mm_numeric" in n or "class" in n:
            agg[(n[:60], r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-60s vgpr %s lds %s grid %s wg %s  calls %d  mean %.3f ms  min %.3f" % (k + (len(v), sum(v) / len(v), min(v))))
PY
done 2>&1 | tee $O/summary.txt
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE"; do
  i=$((${i:-0}+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $PMC -d $OLDPWD/$O/pmc$i -o pmc --output-format csv -- python $OLDPWD/bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 0 --no-pmc --no-other-configs ) > $O/pmc$i.log 2>&1
done
python3 - $O <<'PY' | tee -a $O/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
print("== counters per dispatch (default build), by kernel and grid size")
for f in sorted(glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f, errors="replace")):
        try:
            k = r["Kernel_Name"].split("(")[0][:48]
            if "mm_numeric" not in k:
                continue
            k = (k, r.get("Grid_Size"), r.get("LDS_Block_Size"))
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        except (KeyError, ValueError, TypeError):
            continue
    for k in sorted(agg, key=lambda x: str(x)):
        print("  %-48s grid %-9s lds %-6s " % k + "  ".join("%s=%.4g" % (c, agg[k][c] / cnt[(k, c)]) for c in sorted(agg[k])))
PY
find $O -name "*.csv" -size +1M -delete
