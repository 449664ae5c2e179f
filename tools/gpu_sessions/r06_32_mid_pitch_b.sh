#!/bin/bash
# round 6, GPU session 32: only B's slab pitch changed (KSL + 2), A copied as it lies -- separates the two changes of session 31 (engine kernel only)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s32; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"1,33","fill":0.2},{"mix":"1,36","fill":0.2},{"mix":"1,40","fill":0.2},{"mix":"1,48","fill":0.2},{"mix_m":"1,48","mix_n":"1,36","mix_k":"1,23","fill":0.2}]'
timeout 400 python tools/block_bench.py --size 16384 --label pitchB10 --check --batch "$B" 2>&1 | grep -v "$F" > $O/slab.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_s32/slab.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["label"], d.get("mix_m"), d.get("mix_n"), d.get("mix_k"), d.get("fill"), d.get("kernel"), "kernel_ms", d.get("kernel_ms"), "frac", d.get("frac_of_peak_kernel"), "diff", (d.get("check") or {}).get("max_abs_diff_over_max_abs"), d.get("error"))
PY
