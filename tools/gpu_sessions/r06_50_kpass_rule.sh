#!/bin/bash
# round 6, GPU session 50: the k-pass rule with its second condition (>= 48 products per C block expected): the tests that touch k passes, session 49's lines with the automatic
# choice, config 5 (must keep its four passes) through bench.py
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s50; mkdir -p $O; rm -f $O/*.jsonl
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1200 python -m pytest tests/test_gpu_multiply.py tests/test_gpu_native_multiply.py tests/test_gpu_plan_reuse.py -q -m gpu -x 2>&1 | grep -v "$F" | tail -3
B='[{"mix":"1,32","fill":0.1,"size":45600},{"mix":"1,31","fill":0.1,"size":44175},{"mix":"1,32","fill":0.15,"size":45600},{"mix":"1,32","fill":0.2,"size":32768},{"mix":"1,23","fill":0.2,"size":32775},{"mix":"1,28","fill":0.15,"size":39900},{"mix":"1,23","fill":0.4,"size":16384}]'
timeout 900 python tools/block_bench.py --label auto --batch "$B" 2>&1 | grep -v "$F" >> $O/k.jsonl
python3 - <<'PY'
import json
for r in [json.loads(l) for l in open("gpurun_out/r06_s50/k.jsonl") if l.startswith("{")]:
    if "error" in r: print(r); continue
    print("%-6s %-8s fill %.2f size %6d  passes %d  step_ms %8.3f  TFLOP/s(step) %6.2f" % (r["label"], ",".join(map(str, r["mix_m"])), r["fill"], r["size"], r["k_passes"], r["ms_per_step"], r["tflops_step"]))
PY
timeout 900 python bench.py --workload config5_131072_32x32_fill20_fp32 --steps 2 --warmup 1 --no-pmc --cpu-seconds 0 --no-other-configs 2>/dev/null | grep '^{"metric"' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5 ms_per_step', d['ms_per_step'], 'k_passes', d['config'].get('k_passes'), 'value', d['value'])"
