#!/bin/bash
# round 6, GPU session 40: atomic blocking of water with a DZVP basis -- two blocks of 5 per block of 13 -- and other mixes of small sizes in the benchmark's
# structure (about 1425 block rows, fill 0.1): which kernels serve them and at what rate
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s40; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"2,5,1,13","fill":0.1,"size":10925},{"mix":"1,5,1,13","fill":0.1,"size":12825},{"mix":"1,4,1,8","fill":0.1,"size":8550},{"mix":"1,13","fill":0.1,"size":18525},{"mix":"1,5","fill":0.1,"size":7125},{"mix":"2,5,1,13","fill":0.3,"size":10925},{"mix":"1,9,1,13,1,5","fill":0.1,"size":12825}]'
DBCSR_AMD_MM_VERBOSE=1 timeout 900 python tools/block_bench.py --label mixes --check --batch "$B" 2>&1 | grep -v "$F" > $O/mixes.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_s40/mixes.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        if "error" in d:
            print(d); continue
        print("%-14s fill %.2f  %-100s kernel_ms %8.3f step_ms %8.3f TFLOP/s %6.2f products %9d check %s" % (d["mix_m"], d["fill"], d["kernel"][:100], d["kernel_ms"], d["ms_per_step"], d["tflops_kernel"], d["nproducts"],
              (d.get("check") or {}).get("max_abs_diff_over_max_abs")))
PY
