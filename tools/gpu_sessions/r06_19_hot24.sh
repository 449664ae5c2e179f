#!/bin/bash
# round 6, GPU session 19: hot<24,24,24> with B's columns staged at a pitch of 26 doubles (shipping build) against 24 (lab build, 2-way bank conflicts)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s19; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"1,24","fill":0.1,"size":32768},{"mix":"1,24","fill":0.05,"size":32768},{"mix":"1,16","fill":0.1,"size":16384}]'
for i in 1 2; do
timeout 300 python tools/block_bench.py --size 32768 --label pitch26 --check --batch "$B" 2>&1 | grep -v "$F" >> $O/ship.jsonl
timeout 300 python tools/block_bench.py --size 32768 --label pitch24 --lab --check --batch "$B" 2>&1 | grep -v "$F" >> $O/lab.jsonl
done
python3 - <<'PY'
import json
for f in ("ship", "lab"):
    for l in open("gpurun_out/r06_s19/%s.jsonl" % f):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["label"], d.get("mix_m"), d.get("fill"), d.get("kernel"), "kernel_ms", d.get("kernel_ms"), "frac", d.get("frac_of_peak_kernel"), "diff", (d.get("check") or {}).get("max_abs_diff_over_max_abs"), d.get("error"))
PY
