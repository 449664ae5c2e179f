#!/bin/bash
# round 6, GPU session 37: the one-tile kernel for multiplies whose block dimensions are all <= 8 (mm_numeric_f64_small.h): parity (tests/test_gpu_small_blocks.py, the random
# sweep), then uniform sizes 4 ... 9 in the benchmark's structure (1425 block rows, fill 0.1) with 8 / 4 products in flight and through the kernel that served them before
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s37; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_small_blocks.py -q -m gpu -x 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest.txt
DBCSR_AMD_SWEEP_PLAIN=600 DBCSR_AMD_SWEEP_FORCED=200 timeout 900 python -m pytest tests/test_gpu_random_sweep.py -q -m gpu -x -n 4 -k "matches_oracle or forced" 2>&1 | grep -v "$F" | tail -3 | tee -a $O/pytest.txt
for SW in 2 3 4; do
  B=$(python3 -c 'import json; print(json.dumps([{"mix": "1,%d" % s, "fill": 0.1, "size": 1425 * s} for s in (5, 6, 7, 8)] + [{"mix_m": "1,5", "mix_n": "1,8", "mix_k": "1,5,1,8", "fill": 0.1, "size": 8000}]))')
  DBCSR_AMD_MM_SMALL=$SW timeout 600 python tools/block_bench.py --label small$SW --check --batch "$B" 2>&1 | grep -v "$F" >> $O/sizes.jsonl
done
B=$(python3 -c 'import json; print(json.dumps([{"mix": "1,%d" % s, "fill": 0.1, "size": 1425 * s} for s in (5, 8)]))')
DBCSR_AMD_MM_SMALL=4 DBCSR_AMD_MM_WORK=0 timeout 600 python tools/block_bench.py --label small4_nowork --check --batch "$B" 2>&1 | grep -v "$F" >> $O/sizes.jsonl
python3 - <<'PY'
import json
print("# label   m n k                       kernel                          kernel_ms  TFLOP/s  check")
for l in open("gpurun_out/r06_s37/sizes.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        if "error" in d:
            print(d); continue
        print("%-13s %-26s %-32s %8.3f %8.2f  %s" % (d["label"], "%s %s %s" % (d["mix_m"], d["mix_n"], d["mix_k"]), d["kernel"][:32], d["kernel_ms"], d["tflops_kernel"],
                                                  (d.get("check") or {}).get("max_abs_diff_over_max_abs")))
PY
