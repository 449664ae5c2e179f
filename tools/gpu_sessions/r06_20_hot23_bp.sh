#!/bin/bash
# round 6, GPU session 20 (timing only): hot<23,23,23> with the B fragment reads at a conflict-free pitch of 25 doubles -- the lab build reads there without
# staging there, so its results are wrong -- against the shipping kernel: what are conflict-free B reads worth on config 2?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s20; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"1,23","fill":0.1,"size":32768},{"mix":"1,23","fill":0.05,"size":32768}]'
for i in 1 2; do
timeout 300 python tools/block_bench.py --size 32768 --label ship --batch "$B" 2>&1 | grep -v "$F" >> $O/ship.jsonl
timeout 300 python tools/block_bench.py --size 32768 --label lab_wrong_pitch --lab --batch "$B" 2>&1 | grep -v "$F" >> $O/lab.jsonl
done
python3 - <<'PY'
import json
for f in ("ship", "lab"):
    for l in open("gpurun_out/r06_s20/%s.jsonl" % f):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["label"], d.get("mix_m"), d.get("fill"), d.get("kernel"), "kernel_ms", d.get("kernel_ms"), "frac", d.get("frac_of_peak_kernel"), d.get("error"))
PY
