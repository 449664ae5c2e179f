#!/bin/bash
# round 6, GPU session 42: rows dealt to the XCDs class by class (class_row_deal): parity of the class paths, then the mixes of session 41 and config 3
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s42; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1200 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_multiply.py tests/test_gpu_small_blocks.py -q -m gpu -x 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest.txt
DBCSR_AMD_SWEEP_PLAIN=800 DBCSR_AMD_SWEEP_FORCED=400 timeout 900 python -m pytest tests/test_gpu_random_sweep.py -q -m gpu -x -n 4 -k "matches_oracle or forced" 2>&1 | grep -v "$F" | tail -3 | tee -a $O/pytest.txt
B='[{"mix":"1,5,1,13","fill":0.1,"size":12816},{"mix":"1,5,1,13","fill":0.1,"size":12825},{"mix":"2,5,1,13","fill":0.1,"size":10925},{"mix":"1,13,1,23","fill":0.1,"size":25650},{"mix":"1,5,1,13,1,5,1,23","fill":0.1,"size":16400},{"mix":"1,5,1,13","fill":0.1,"size":12825,"env":["DBCSR_AMD_MM_CLASSES=0"]},{"mix":"1,13,1,23","fill":0.1,"size":25650,"env":["DBCSR_AMD_MM_CLASSES=0"]}]'
timeout 900 python tools/block_bench.py --label deal --check --batch "$B" 2>&1 | grep -v "$F" > $O/mixes.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_s42/mixes.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        if "error" in d:
            print(d); continue
        print("%-26s size %6d %s %-60s kernel_ms %8.3f TFLOP/s %6.2f check %s" % (d["mix_m"], d["size"], d["env"], d["kernel"][:60], d["kernel_ms"], d["tflops_kernel"], (d.get("check") or {}).get("max_abs_diff_over_max_abs")))
PY
timeout 600 python bench.py --workload config3_32768_mixed13_23_32_fill5_fp64 --steps 5 --warmup 2 --no-pmc --cpu-seconds 0 2>/dev/null | grep '^{"metric"' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config3 ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), d['roofline'].get('kernel','')[:60])"
