#!/bin/bash
# round 6, GPU session 64: very sparse mixes of small sizes through the class kernels with G C blocks per wave (the lab build's DBCSR_AMD_MM_CLASS_G): is the wave
# start rate the bound there too?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s64; mkdir -p $O; rm -f $O/*.jsonl
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"1,5,1,13","fill":0.01,"size":40000},{"mix":"1,13,1,23,1,32","fill":0.01,"size":65536},{"mix":"1,13","fill":0.01,"size":74000}]'
for G in 1 4 8; do
  DBCSR_AMD_MM_CLASS_G=$G timeout 600 python tools/block_bench.py --lab --label G$G --check --batch "$B" 2>&1 | grep -v "$F" >> $O/g.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_s64/g.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        if "error" in d: print(d); continue
        print("%-4s %-22s fill %.2f %-46s kernel_ms %8.3f C blocks %9d products per block %.1f check %s" % (d["label"], d["mix_m"], d["fill"], d["kernel"][:46], d["kernel_ms"], d["c_nblks"], d["products_per_c_block"], (d.get("check") or {}).get("max_abs_diff_over_max_abs")))
PY
