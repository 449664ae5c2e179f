#!/bin/bash
# round 6, GPU session 54: the filtered multiply (no plan reuse possible: the product lists depend on the norms) on BASELINE's one-GPU shapes with this round's kernels --
# unfiltered / a filter that drops nothing / a filter that drops blocks (tools/filtered_multiply_timing.py, as profiles/r02_filtered_sparse_multiply.txt), and its kernel trace
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=$PWD/gpurun_out/r06_s54; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for S in config2 config3 config4; do
  echo "# SHAPE=$S" >> $O/filtered.txt
  SHAPE=$S timeout 600 python tools/filtered_multiply_timing.py auto 2>&1 | grep -v "$F" >> $O/filtered.txt
done
cat $O/filtered.txt
export TMPDIR=/tmp
( cd /tmp && SHAPE=config2 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/filtered_multiply_timing.py auto ) > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "bitmap_from_index" in r["Kernel_Name"])
while idx > 0 and "bitmap_from_index" in rows[idx - 1]["Kernel_Name"]:
    idx -= 1
t0 = int(rows[idx]["Start_Timestamp"])
print("# config 2's shape, the LAST multiply of the run (filter_eps = 500): start us, duration us, kernel (>= 20 us)")
busy = 0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if e - s >= 20000:
        print("%9.1f %9.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:80]))
print("# span %.1f us, kernels busy %.1f us, %d launches" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e3, busy / 1e3, len(rows) - idx))
PY
