#!/bin/bash
# round 6, GPU session 43: uniform RECTANGULAR triplets in the benchmark's structure (1425 block rows / columns / inner blocks, fill 0.1): which kernel serves them, at what rate
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s43; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B=$(python3 - <<'PY'
import json
T = [(5, 13, 23), (23, 5, 13), (13, 13, 5), (13, 5, 13), (5, 5, 13), (32, 32, 8), (8, 8, 32), (4, 4, 32), (4, 4, 13), (8, 8, 13), (6, 6, 23), (23, 23, 5), (23, 23, 13), (13, 13, 23), (16, 32, 16), (32, 16, 32), (24, 24, 8), (9, 9, 32), (32, 9, 9)]
print(json.dumps([{"mix_m": "1,%d" % m, "mix_n": "1,%d" % n, "mix_k": "1,%d" % k, "fill": 0.1, "size": 1425 * max(m, n, k)} for m, n, k in T]))
PY
)
timeout 1500 python tools/block_bench.py --label rect --check --batch "$B" 2>&1 | grep -v "$F" > $O/rect.jsonl
python3 - <<'PY'
import json
print("# m x n x k        kernel                                                       kernel_ms  TFLOP/s  products  products per C block  check")
for l in open("gpurun_out/r06_s43/rect.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        if "error" in d:
            print(d); continue
        print("%2d x %2d x %2d    %-60s %8.3f %8.2f %9d %6.1f  %s" % (d["mix_m"][1], d["mix_n"][1], d["mix_k"][1], d["kernel"][:60], d["kernel_ms"], d["tflops_kernel"], d["nproducts"], d["products_per_c_block"],
              (d.get("check") or {}).get("max_abs_diff_over_max_abs")))
PY
