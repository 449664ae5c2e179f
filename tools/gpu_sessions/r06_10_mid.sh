#!/bin/bash
# round 6, GPU session 10: blocks of 33 ... 40 through the one-wave kernel mm_numeric_f64_mid: parity, then block_bench against the
# workgroup kernel (DBCSR_AMD_MM_MID=0) on the same box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s10; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 900 python -m pytest tests/test_gpu_big_blocks.py tests/test_gpu_libsmm.py -q -x 2>&1 | grep -v "$F" | tail -12 ) > $O/pytest_big.txt 2>&1
tail -6 $O/pytest_big.txt
DBCSR_AMD_SWEEP_BIG=60 DBCSR_AMD_SWEEP_MID=120 timeout 600 python -m pytest tests/test_gpu_random_sweep.py -q -x -k "large_blocks or 33_to_40" 2>&1 | grep -v "$F" | tail -6 > $O/pytest_sweep.txt
tail -3 $O/pytest_sweep.txt
B='[{"mix":"1,33","fill":0.2},{"mix":"1,36","fill":0.2},{"mix":"1,37","fill":0.2},{"mix":"1,40","fill":0.2},{"mix_m":"1,36","mix_n":"1,40","mix_k":"1,23","fill":0.2},{"mix":"1,33","fill":0.05},{"mix":"1,40","fill":0.5,"size":8192}]'
timeout 400 python tools/block_bench.py --size 16384 --label mid --check --batch "$B" 2>&1 | grep -v "$F" > $O/mid.jsonl
DBCSR_AMD_MM_MID=0 timeout 400 python tools/block_bench.py --size 16384 --label big --check --batch "$B" 2>&1 | grep -v "$F" > $O/big.jsonl
python3 - <<'PY'
import json
for f in ("gpurun_out/r06_s10/mid.jsonl", "gpurun_out/r06_s10/big.jsonl"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["label"], d["mix_m"], d["mix_n"], d["mix_k"], d["fill"], d["kernel"], "kernel_ms", d["kernel_ms"], "frac", d["frac_of_peak_kernel"], "check", d.get("check"))
PY
