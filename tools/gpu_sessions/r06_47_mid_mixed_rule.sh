#!/bin/bash
# round 6, GPU session 47: mixes around 32 without a dominant size -- the slab kernel (largest size exact, the others through its second launch) against the workgroup kernel
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s47; mkdir -p $O; rm -f $O/*.jsonl
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"1,30,1,36","fill":0.1,"size":32768},{"mix":"1,33,1,36","fill":0.1,"size":32768},{"mix":"1,30,1,40","fill":0.1,"size":32768},{"mix":"1,28,1,36","fill":0.1,"size":32768},{"mix":"1,34,1,40","fill":0.1,"size":32768},{"mix":"1,33,1,48,1,41,1,30","fill":0.1,"size":32768},{"mix":"1,23,1,36","fill":0.1,"size":32768},{"mix":"3,36,1,30","fill":0.1,"size":32768},{"mix":"1,36,3,30","fill":0.1,"size":32768},{"mix":"1,13,1,36","fill":0.1,"size":24000}]'
timeout 900 python tools/block_bench.py --label mid --batch "$B" 2>&1 | grep -v "$F" > $O/mid.jsonl
DBCSR_AMD_MM_MID=0 timeout 900 python tools/block_bench.py --label big --batch "$B" 2>&1 | grep -v "$F" > $O/big.jsonl
python3 - <<'PY'
import json
def load(f):
    return [json.loads(l) for l in open(f) if l.startswith("{")]
d, g = load("gpurun_out/r06_s47/mid.jsonl"), load("gpurun_out/r06_s47/big.jsonl")
for a, b in zip(d, g):
    print("%-24s %-28s %8.3f | %-28s %8.3f | ratio %.2f" % (",".join(map(str, a["mix_m"])), a["kernel"][:28], a["kernel_ms"], b["kernel"][:28], b["kernel_ms"], b["kernel_ms"] / a["kernel_ms"]))
PY
