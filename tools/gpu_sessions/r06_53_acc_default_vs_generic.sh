#!/bin/bash
# round 6, GPU session 53: the acc ABI, sixteen streams x 30000-entry stacks: the library's choice against its run-time-size stack kernel (DBCSR_AMD_SMM_EXACT=0
# DBCSR_AMD_SMM_MID=0) over a spread of triplets -- a line where the choice loses is a rule to look at
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s53; mkdir -p $O; rm -f $O/*.txt
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for T in "9 9 9" "13 13 13" "16 16 16" "23 23 23" "32 32 32" "5 13 23" "23 5 13" "13 13 5" "32 9 9" "9 32 9" "9 9 32" "24 24 8" "8 8 24" "16 32 16" "26 26 26" "13 23 32" "32 23 13" "4 4 32" "33 33 33" "36 36 23" "40 23 40" "45 45 45" "64 64 64"; do
  set -- $T
  a=$(timeout 120 python tools/acc_bench.py 20 30000 $1 $2 $3 4000 10000 10000 --threads 16 2>&1 | grep -v "$F" | tail -1)
  b=$(DBCSR_AMD_SMM_EXACT=0 DBCSR_AMD_SMM_MID=0 timeout 120 python tools/acc_bench.py 20 30000 $1 $2 $3 4000 10000 10000 --threads 16 2>&1 | grep -v "$F" | tail -1)
  echo "$a" >> $O/default.txt; echo "$b" >> $O/generic.txt
done
python3 - <<'PY'
import re
def load(f):
    out = []
    for l in open(f):
        m = re.search(r"m=(\d+) n=(\d+) k=(\d+).*?: ([\d.]+) GFLOP/s.*\[(.*)\]", l)
        out.append((m.group(1), m.group(2), m.group(3), float(m.group(4)), m.group(5)) if m else None)
    return out
for a, b in zip(load("gpurun_out/r06_s53/default.txt"), load("gpurun_out/r06_s53/generic.txt")):
    if not a or not b:
        print("?", a, b); continue
    flag = "  <-- LOSES" if b[3] > 1.03 * a[3] else ""
    print("%3s x %3s x %3s  %9.1f GFLOP/s  %-46s | %9.1f  %-40s | %.2f%s" % (a[0], a[1], a[2], a[3], a[4][:46], b[3], b[4][:40], a[3] / b[3], flag))
PY
