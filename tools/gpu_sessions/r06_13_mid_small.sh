#!/bin/bash
# round 6, GPU session 13 (VERDICT r05 item 2, config 3's regime): the one-wave SLAB kernel (mm_numeric_f64_mid: 6-9 KB of LDS per wave, any inner dimension)
# on blocks of 32 / 23 / 13 at few products per C block against the exact-size kernels that stage whole blocks (hot<S,S,S>: 9.5-18 KB per wave)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s13; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"1,32","fill":0.05},{"mix":"1,23","fill":0.05},{"mix":"1,13","fill":0.05},{"mix":"1,32","fill":0.1},{"mix":"1,23","fill":0.1}]'
timeout 300 python tools/block_bench.py --size 32768 --label exact --lab --check --batch "$B" 2>&1 | grep -v "$F" > $O/exact.jsonl
timeout 300 python tools/block_bench.py --size 32768 --label slab8 --lab --env DBCSR_AMD_MM_MID=8 --check --batch "$B" 2>&1 | grep -v "$F" > $O/slab8.jsonl
timeout 300 python tools/block_bench.py --size 32768 --label slab16 --lab --env DBCSR_AMD_MM_MID=16 --check --batch "$B" 2>&1 | grep -v "$F" > $O/slab16.jsonl
python3 - <<'PY'
import json
for f in ("exact", "slab8", "slab16"):
    for l in open("gpurun_out/r06_s13/%s.jsonl" % f):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["label"], d.get("mix_m"), d.get("fill"), d.get("kernel"), "kernel_ms", d.get("kernel_ms"), "frac", d.get("frac_of_peak_kernel"), "diff", (d.get("check") or {}).get("max_abs_diff_over_max_abs"), d.get("error"))
PY
