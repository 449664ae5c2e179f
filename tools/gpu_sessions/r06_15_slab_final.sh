#!/bin/bash
# round 6, GPU session 15: the slab kernel with its final rules (8 units and more; (32, 23) / (23, 32) classes; straight C stores): parity, the block sizes again,
# config 3 with DBCSR_AMD_MM_MID = 0 / 3 / default, the kernel trace of config 3 per launch
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s15; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 1500 python -m pytest tests/test_gpu_big_blocks.py tests/test_gpu_kernel_variants.py tests/test_gpu_multiply.py tests/test_gpu_plan_reuse.py tests/test_gpu_native_multiply.py -q -x 2>&1 | grep -v "$F" | tail -12 ) > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
DBCSR_AMD_SWEEP_BIG=60 DBCSR_AMD_SWEEP_MID=120 timeout 900 python -m pytest tests/test_gpu_random_sweep.py -q -x 2>&1 | grep -v "$F" | tail -5 > $O/pytest_sweep.txt
tail -2 $O/pytest_sweep.txt
B='[{"mix":"1,32","fill":0.05,"size":32768},{"mix":"1,32","fill":0.1,"size":32768},{"mix":"1,30","fill":0.1,"size":32768},{"mix":"1,29","fill":0.1,"size":32768},{"mix":"1,33","fill":0.2},{"mix":"1,36","fill":0.2},{"mix":"1,40","fill":0.2},{"mix_m":"1,40","mix_n":"1,28","mix_k":"1,32","fill":0.2}]'
timeout 400 python tools/block_bench.py --size 16384 --label slab --check --batch "$B" 2>&1 | grep -v "$F" > $O/slab.jsonl
DBCSR_AMD_MM_MID=0 timeout 400 python tools/block_bench.py --size 16384 --label exact --batch "$B" 2>&1 | grep -v "$F" > $O/exact.jsonl
python3 - <<'PY'
import json
for f in ("slab", "exact"):
    for l in open("gpurun_out/r06_s15/%s.jsonl" % f):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["label"], d.get("mix_m"), d.get("mix_n"), d.get("mix_k"), d.get("fill"), d.get("kernel"), "kernel_ms", d.get("kernel_ms"), "frac", d.get("frac_of_peak_kernel"), "diff", (d.get("check") or {}).get("max_abs_diff_over_max_abs"), d.get("error"))
PY
for M in 0 3 1 0 3 1; do
  ( export DBCSR_AMD_MM_MID=$M; timeout 300 python -c "
import json, bench
r = bench.run_other_config('config3_32768_mixed13_23_32_fill5_fp64', steps=5)
print('config3 DBCSR_AMD_MM_MID=$M', r.get('ms_per_step'), r.get('value'))" 2>/dev/null | grep config3 ) >> $O/config3.txt
done
cat $O/config3.txt
