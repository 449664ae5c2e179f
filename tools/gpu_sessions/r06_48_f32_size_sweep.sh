#!/bin/bash
# round 6, GPU session 48: fp32, uniform block sizes 4 ... 32 in the benchmark's structure (1425 block rows, fill 0.1) and a few mixes: kernel chosen, rate
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s48; mkdir -p $O; rm -f $O/*.jsonl
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B=$(python3 -c 'import json; print(json.dumps([{"mix": "1,%d" % s, "fill": 0.1, "size": 1425 * s, "dtype": "f32"} for s in (4, 5, 8, 9, 12, 13, 16, 17, 20, 23, 24, 25, 28, 31, 32)] + [{"mix": m, "fill": 0.1, "size": z, "dtype": "f32"} for m, z in (("1,13,1,23,1,32", 32768), ("1,5,1,13", 12816), ("2,5,1,13", 10925), ("1,36", 32768), ("1,40", 32768))]))')
timeout 1500 python tools/block_bench.py --label f32 --check --batch "$B" 2>&1 | grep -v "$F" > $O/f32.jsonl
python3 - <<'PY'
import json
print("# mix            kernel                                    kernel_ms  TFLOP/s  frac of the 157.3 fp32 peak   check")
for l in open("gpurun_out/r06_s48/f32.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        if "error" in d:
            print(d); continue
        print("%-16s %-44s %8.3f %8.2f %8.3f   %s" % (",".join(map(str, d["mix_m"])), d["kernel"][:44], d["kernel_ms"], d["tflops_kernel"], d["tflops_kernel"] / 157.3, (d.get("check") or {}).get("max_abs_diff_over_max_abs")))
PY
