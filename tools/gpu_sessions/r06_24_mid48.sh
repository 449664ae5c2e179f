#!/bin/bash
# round 6, GPU session 24: the one-wave slab kernel on blocks of 41 ... 48 (11 / 12 units of 4 x 4) against the workgroup kernel (DBCSR_AMD_MM_MID=0)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s24; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 900 python -m pytest tests/test_gpu_big_blocks.py -q -x 2>&1 | grep -v "$F" | tail -5 ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
DBCSR_AMD_SWEEP_FORCED=0 DBCSR_AMD_SWEEP_PLAIN=0 DBCSR_AMD_SWEEP_BIG=100 DBCSR_AMD_SWEEP_MID=400 timeout 900 python -m pytest tests/test_gpu_random_sweep.py -q -x -n 4 2>&1 | grep -v "$F" | tail -3
B='[{"mix":"1,41","fill":0.2},{"mix":"1,44","fill":0.2},{"mix":"1,45","fill":0.2},{"mix":"1,48","fill":0.2},{"mix":"1,48","fill":0.05},{"mix_m":"1,48","mix_n":"1,36","mix_k":"1,23","fill":0.2},{"mix":"1,40","fill":0.2}]'
timeout 400 python tools/block_bench.py --size 16384 --label slab --check --batch "$B" 2>&1 | grep -v "$F" > $O/slab.jsonl
DBCSR_AMD_MM_MID=0 timeout 400 python tools/block_bench.py --size 16384 --label workgroup --batch "$B" 2>&1 | grep -v "$F" > $O/big.jsonl
python3 - <<'PY'
import json
for f in ("slab", "big"):
    for l in open("gpurun_out/r06_s24/%s.jsonl" % f):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["label"], d.get("mix_m"), d.get("mix_n"), d.get("mix_k"), d.get("fill"), d.get("kernel"), "kernel_ms", d.get("kernel_ms"), "frac", d.get("frac_of_peak_kernel"), "diff", (d.get("check") or {}).get("max_abs_diff_over_max_abs"), d.get("error"))
PY
