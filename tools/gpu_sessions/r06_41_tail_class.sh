#!/bin/bash
# round 6, GPU session 41: the 5 / 13 mix with and without a tail block (a third size with a handful of blocks): 10.5 against 4.4 ms in session 40 -- where does it go?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s41; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"1,5,1,13","fill":0.1,"size":12816},{"mix":"1,5,1,13","fill":0.1,"size":12825},{"mix":"2,5,1,13","fill":0.1,"size":10925},{"mix":"2,5,1,13","fill":0.1,"size":10929},{"mix":"1,5,1,13","fill":0.1,"size":12825,"env":["DBCSR_AMD_MM_CLASSES=0"]}]'
timeout 900 python tools/block_bench.py --label tail --batch "$B" 2>&1 | grep -v "$F" > $O/mixes.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_s41/mixes.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        if "error" in d:
            print(d); continue
        print("%-14s size %6d %s %-100s kernel_ms %8.3f step_ms %8.3f GFLOP %7.1f products %9d c_nblks %d" % (d["mix_m"], d["size"], d["env"], d["kernel"][:100], d["kernel_ms"], d["ms_per_step"], d["flop"] / 1e9, d["nproducts"], d["c_nblks"]))
PY
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/block_bench.py --label tail --mix 1,5,1,13 --size 12825 --fill 0.1 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "mm_numeric" in r["Kernel_Name"]]
# last multiply: the last 10 launches
for r in rows[-12:]:
    print("%-60s grid %9s wg %4s lds %6s  %9.3f ms" % (r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("LDS_Block_Size", "?"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
