#!/bin/bash
# round 6, GPU session 11: the 9 x 9 variant of mm_numeric_f64_mid (blocks of 33 ... 36): two operand sets at three waves per SIMD (shipping build)
# against one set at four waves per SIMD (lab build: 128 registers, one of them spilled)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s11; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"1,33","fill":0.2},{"mix":"1,36","fill":0.2},{"mix":"1,34","fill":0.1}]'
timeout 400 python tools/block_bench.py --size 16384 --label ship_2sets --check --batch "$B" 2>&1 | grep -v "$F" > $O/ship.jsonl
timeout 400 python tools/block_bench.py --size 16384 --label lab_4waves --lab --check --batch "$B" 2>&1 | grep -v "$F" > $O/lab.jsonl
timeout 400 python tools/block_bench.py --size 16384 --label ship_2sets_again --batch "$B" 2>&1 | grep -v "$F" >> $O/ship.jsonl
python3 - <<'PY'
import json
for f in ("gpurun_out/r06_s11/ship.jsonl", "gpurun_out/r06_s11/lab.jsonl"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["label"], d.get("mix_m"), d.get("fill"), d.get("kernel"), "kernel_ms", d.get("kernel_ms"), "frac", d.get("frac_of_peak_kernel"), d.get("error"))
PY
