#!/bin/bash
# round 6, GPU session 51: the step WITHOUT plan reuse (DBCSR_AMD_MM_PLAN=0: symbolic phase, product lists and launch order rebuilt in every multiply -- what a host pays
# whose sparsity patterns change from one multiply to the next) against the step with it, BASELINE's one-GPU shapes and two mixes
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s51; mkdir -p $O; rm -f $O/*.jsonl
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"1,4","fill":0.1,"size":4096},{"mix":"1,23","fill":0.1,"size":32768},{"mix":"1,13,1,23,1,32","fill":0.05,"size":32768},{"mix":"1,23","fill":0.01,"size":131072},{"mix":"1,5,1,13","fill":0.1,"size":12816},{"mix":"1,5","fill":0.1,"size":7125},{"mix":"1,36","fill":0.1,"size":32768}]'
timeout 900 python tools/block_bench.py --label warm --batch "$B" 2>&1 | grep -v "$F" >> $O/p.jsonl
DBCSR_AMD_MM_PLAN=0 timeout 900 python tools/block_bench.py --label cold --batch "$B" 2>&1 | grep -v "$F" >> $O/p.jsonl
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r06_s51/p.jsonl") if l.startswith("{")]
w = [r for r in rows if r["label"] == "warm"]; c = [r for r in rows if r["label"] == "cold"]
for a, b in zip(w, c):
    print("%-16s fill %.2f size %6d  kernel_ms %8.3f  step warm %8.3f  step cold %8.3f  (+%.2f ms, %.0f %%)  C blocks %9d products %9d" % (",".join(map(str, a["mix_m"])), a["fill"], a["size"], a["kernel_ms"], a["ms_per_step"], b["ms_per_step"],
          b["ms_per_step"] - a["ms_per_step"], 100 * (b["ms_per_step"] / a["ms_per_step"] - 1), a["c_nblks"], a["nproducts"]))
PY
