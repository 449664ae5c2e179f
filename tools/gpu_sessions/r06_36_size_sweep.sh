#!/bin/bash
# round 6, GPU session 36: uniform block sizes 4 ... 32, the benchmark's structure (1425 block rows, fill 0.1: 29 M products, 14.3 per C block) at every size -- the
# fraction of the fp64 peak per size: where the exact-size kernels stand against each other (the 8-way bank conflict at 32 was found by accident: look at all of them)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s36; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B=$(python3 -c 'import json; print(json.dumps([{"mix": "1,%d" % s, "fill": 0.1, "size": 1425 * s} for s in range(4, 33)]))')
timeout 1500 python tools/block_bench.py --label sizes --batch "$B" 2>&1 | grep -v "$F" > $O/sizes.jsonl
python3 - <<'PY'
import json
print("# size  kernel                          kernel_ms  TFLOP/s  frac_of_fp64_peak  useful/issued MACs (tiles of 8 x 8, k in fours)")
for l in open("gpurun_out/r06_s36/sizes.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        if "error" in d:
            print(d); continue
        s = d["mix_m"][1]
        pad = (8 * ((s + 7) // 8)) ** 2 * 4 * ((s + 3) // 4)
        print("%5d  %-32s %8.3f %8.2f %10.3f %14.2f" % (s, d["kernel"], d["kernel_ms"], d["tflops_kernel"], d["frac_of_peak_kernel"], s ** 3 / pad))
PY
