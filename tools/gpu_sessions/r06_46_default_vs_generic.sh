#!/bin/bash
# round 6, GPU session 46: the engine's own choice against its run-time-size kernels (DBCSR_AMD_MM_HOT=0 CLASSES=0 SMALL=0 MID=0) on a spread of workloads: a line where
# the choice LOSES is a dispatch rule to look at (sessions 40-44 found two that way)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s46; mkdir -p $O; rm -f $O/*.jsonl
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B=$(python3 - <<'PY'
import json
W = [("1,23", 0.01, 65536), ("1,23", 0.3, 16384), ("1,13,1,23,1,32", 0.01, 65536), ("1,13,1,23,1,32", 0.2, 16384), ("1,5,1,13", 0.01, 40000), ("1,5,1,13", 0.3, 8000),
     ("3,5,1,13,2,9", 0.1, 12000), ("1,16", 0.1, 22800), ("1,12,1,20", 0.1, 22800), ("1,26,1,32", 0.1, 32768), ("1,7,1,23", 0.1, 16384), ("1,4,1,23", 0.1, 16384),
     ("1,30,1,36", 0.1, 32768), ("1,23,1,40", 0.1, 32768), ("1,8,1,16,1,24,1,32", 0.1, 24000), ("1,3,1,6,1,9,1,12,1,15", 0.1, 12000), ("1,32", 0.02, 65536), ("1,9", 0.3, 8000)]
print(json.dumps([{"mix": m, "fill": f, "size": s} for m, f, s in W]))
PY
)
timeout 900 python tools/block_bench.py --label default --batch "$B" 2>&1 | grep -v "$F" > $O/default.jsonl
DBCSR_AMD_MM_HOT=0 DBCSR_AMD_MM_CLASSES=0 DBCSR_AMD_MM_SMALL=0 DBCSR_AMD_MM_MID=0 timeout 900 python tools/block_bench.py --label generic --batch "$B" 2>&1 | grep -v "$F" > $O/generic.jsonl
python3 - <<'PY'
import json
def load(f):
    out = []
    for l in open(f):
        if l.startswith("{"):
            out.append(json.loads(l))
    return out
d, g = load("gpurun_out/r06_s46/default.jsonl"), load("gpurun_out/r06_s46/generic.jsonl")
print("# mix, fill, size: the engine's choice (kernel ms, step ms) | run-time-size kernels (kernel ms) | products per C block | ratio generic / default")
for a, b in zip(d, g):
    if "error" in a or "error" in b:
        print(a.get("error"), b.get("error")); continue
    flag = "  <-- LOSES" if b["kernel_ms"] < 0.97 * a["kernel_ms"] else ""
    print("%-24s %.2f %6d  %-58s %8.3f %8.3f | %-24s %8.3f | %6.1f | %.2f%s" % (",".join(map(str, a["mix_m"])), a["fill"], a["size"], a["kernel"][:58], a["kernel_ms"], a["ms_per_step"],
          b["kernel"][:24], b["kernel_ms"], a["products_per_c_block"], b["kernel_ms"] / a["kernel_ms"], flag))
PY
