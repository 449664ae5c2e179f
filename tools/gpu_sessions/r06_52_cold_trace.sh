#!/bin/bash
# round 6, GPU session 52: kernel trace of the multiply WITHOUT plan reuse on config 1's shape and on uniform 5 x 5 blocks: where do the 0.5 / 1.2 ms of the plan go?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=$PWD/gpurun_out/r06_s52; mkdir -p $O
export TMPDIR=/tmp DBCSR_AMD_MM_PLAN=0
for W in "1,4 4096" "1,5 7125"; do
  set -- $W
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$2 -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/block_bench.py --label cold --mix $1 --size $2 --fill 0.1 --steps 5 ) > $O/trace_$2.log 2>&1
  f=$(find $O/trace_$2 -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last multiply: from the last c_bitmap / bitmap_from_index to the end
idx = max(i for i, r in enumerate(rows) if "bitmap_from_index" in r["Kernel_Name"])
while idx > 0 and "bitmap_from_index" in rows[idx - 1]["Kernel_Name"]:
    idx -= 1
t0 = int(rows[idx]["Start_Timestamp"])
busy = 0
print("# one multiply without plan reuse: start us, duration us, kernel")
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:70]))
print("# span %.1f us, kernels busy %.1f us" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e3, busy / 1e3))
PY
done
