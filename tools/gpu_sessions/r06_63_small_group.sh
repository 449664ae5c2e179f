#!/bin/bash
# round 6, GPU session 63: the small-block kernel with G C blocks per wave (the one-block form is bound by the wave start rate: session 62): parity, then G = 1, 2, 4, 8, 16
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s63; mkdir -p $O; rm -f $O/*.jsonl
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for G in 4 3 16; do DBCSR_AMD_MM_SMALL_G=$G timeout 900 python -m pytest tests/test_gpu_small_blocks.py -q -m gpu -x 2>&1 | grep -v "$F" | tail -1; done
DBCSR_AMD_SWEEP_PLAIN=600 DBCSR_AMD_SWEEP_FORCED=200 timeout 900 python -m pytest tests/test_gpu_random_sweep.py -q -m gpu -x -n 4 -k "matches_oracle or forced" 2>&1 | grep -v "$F" | tail -1
B=$(python3 -c 'import json; print(json.dumps([{"mix": "1,%d" % s, "fill": 0.1, "size": 1425 * s} for s in (5, 6, 8)] + [{"mix_m": "1,5", "mix_n": "1,8", "mix_k": "1,5,1,8", "fill": 0.1, "size": 8000}, {"mix": "1,5", "fill": 0.01, "size": 28495}]))')
for G in 1 2 4 8 16; do
  DBCSR_AMD_MM_SMALL_G=$G timeout 600 python tools/block_bench.py --label G$G --check --batch "$B" 2>&1 | grep -v "$F" >> $O/g.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_s63/g.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        if "error" in d: print(d); continue
        print("%-4s %-22s fill %.2f %-26s kernel_ms %8.3f  check %s" % (d["label"], "%s %s %s" % (d["mix_m"], d["mix_n"], d["mix_k"]), d["fill"], d["kernel"][:26], d["kernel_ms"], (d.get("check") or {}).get("max_abs_diff_over_max_abs")))
PY
