#!/bin/bash
# round 6, GPU session 14: the one-wave slab kernel as the kernel of the blocks of 25 ... 40 (work records, C epilogue through LDS) and of config 3's
# (32, 32) class: parity of the suites that meet it, block_bench against the exact-size kernels, config 3 with DBCSR_AMD_MM_MID = 0 / 1 / 2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s14; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 1500 python -m pytest tests/test_gpu_big_blocks.py tests/test_gpu_kernel_variants.py tests/test_gpu_multiply.py tests/test_gpu_plan_reuse.py tests/test_gpu_native_multiply.py -q -x 2>&1 | grep -v "$F" | tail -12 ) > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
DBCSR_AMD_SWEEP_BIG=60 DBCSR_AMD_SWEEP_MID=120 timeout 900 python -m pytest tests/test_gpu_random_sweep.py -q -x 2>&1 | grep -v "$F" | tail -5 > $O/pytest_sweep.txt
tail -3 $O/pytest_sweep.txt
B='[{"mix":"1,32","fill":0.05,"size":32768},{"mix":"1,32","fill":0.1,"size":32768},{"mix":"1,28","fill":0.1,"size":32768},{"mix":"1,25","fill":0.1,"size":32768},{"mix":"1,33","fill":0.2},{"mix":"1,36","fill":0.2},{"mix":"1,40","fill":0.2},{"mix_m":"1,32","mix_n":"1,32","mix_k":"1,23","fill":0.1,"size":32768}]'
timeout 400 python tools/block_bench.py --size 16384 --label slab --check --batch "$B" 2>&1 | grep -v "$F" > $O/slab.jsonl
DBCSR_AMD_MM_MID=0 timeout 400 python tools/block_bench.py --size 16384 --label exact --batch "$B" 2>&1 | grep -v "$F" > $O/exact.jsonl
python3 - <<'PY'
import json
for f in ("slab", "exact"):
    for l in open("gpurun_out/r06_s14/%s.jsonl" % f):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["label"], d.get("mix_m"), d.get("mix_k"), d.get("fill"), d.get("kernel"), "kernel_ms", d.get("kernel_ms"), "frac", d.get("frac_of_peak_kernel"), "diff", (d.get("check") or {}).get("max_abs_diff_over_max_abs"), d.get("error"))
PY
for M in 0 1 2 1 0; do
  ( export DBCSR_AMD_MM_MID=$M; timeout 300 python -c "
import json, bench
r = bench.run_other_config('config3_32768_mixed13_23_32_fill5_fp64', steps=5)
print('config3 DBCSR_AMD_MM_MID=$M', r.get('ms_per_step'), r.get('roofline', {}).get('kernel_ms'), r.get('value'), r.get('roofline', {}).get('frac'), r.get('roofline', {}).get('kernel'))" 2>/dev/null | grep config3 ) >> $O/config3.txt
done
cat $O/config3.txt
