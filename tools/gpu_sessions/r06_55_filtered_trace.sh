#!/bin/bash
# round 6, GPU session 55: kernel statistics of the filtered-multiply tool on config 3's and config 4's shapes (what the filter costs there beyond config 2's breakdown)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=$PWD/gpurun_out/r06_s55; mkdir -p $O
export TMPDIR=/tmp
for S in config3 config4; do
  ( cd /tmp && SHAPE=$S timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$S -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/filtered_multiply_timing.py auto ) > $O/trace_$S.log 2>&1
  f=$(find $O/trace_$S -name "*kernel_trace.csv" | head -1)
  echo "== $S"; grep "^symbolic" $O/trace_$S.log
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split the run into multiplies at bitmap_from_index bursts; report the last multiply of each of the three filter settings (18 multiplies: 6 per setting)
starts = [i for i, r in enumerate(rows) if "bitmap_from_index" in r["Kernel_Name"] and (i == 0 or "bitmap_from_index" not in rows[i - 1]["Kernel_Name"])]
starts.append(len(rows))
mult = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
for label, m in (("eps 0", mult[5]), ("eps 1e-3", mult[11]), ("eps large", mult[17])) if len(mult) >= 18 else []:
    agg = collections.OrderedDict()
    for r in rows[m[0]:m[1]]:
        k = r["Kernel_Name"].split("(")[0][:56]
        agg[k] = agg.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = (int(rows[m[1] - 1]["End_Timestamp"]) - int(rows[m[0]]["Start_Timestamp"])) / 1e6
    print("  %s: span %.2f ms; " % (label, span) + ", ".join("%s %.2f" % (k.replace("dbcsr_amd::", "").replace("void ", ""), v / 1e6) for k, v in agg.items() if v >= 100000))
PY
done
