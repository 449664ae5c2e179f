#!/bin/bash
# round 6, GPU session 16: the exact-size kernel with a padded LDS pitch for blocks of 16 / 32 (hot<32,32,32> met 8-way bank conflicts on its B
# fragment reads) against the slab kernel; config 3 with the slab classes writing C through LDS
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s16; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 1500 python -m pytest tests/test_gpu_big_blocks.py tests/test_gpu_kernel_variants.py tests/test_gpu_multiply.py tests/test_gpu_plan_reuse.py -q -x 2>&1 | grep -v "$F" | tail -12 ) > $O/pytest.txt 2>&1
grep -E "passed|failed" $O/pytest.txt
B='[{"mix":"1,32","fill":0.05,"size":32768},{"mix":"1,32","fill":0.1,"size":32768},{"mix":"1,16","fill":0.1,"size":16384},{"mix":"1,32","fill":0.4,"size":8192}]'
timeout 400 python tools/block_bench.py --size 16384 --label slab --check --batch "$B" 2>&1 | grep -v "$F" > $O/slab.jsonl
DBCSR_AMD_MM_MID=0 timeout 400 python tools/block_bench.py --size 16384 --label hot_padded --check --batch "$B" 2>&1 | grep -v "$F" > $O/exact.jsonl
python3 - <<'PY'
import json
for f in ("slab", "exact"):
    for l in open("gpurun_out/r06_s16/%s.jsonl" % f):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["label"], d.get("mix_m"), d.get("fill"), d.get("size"), d.get("kernel"), "kernel_ms", d.get("kernel_ms"), "frac", d.get("frac_of_peak_kernel"), "diff", (d.get("check") or {}).get("max_abs_diff_over_max_abs"), d.get("error"))
PY
for M in 0 3 1 0 3 1; do
  ( export DBCSR_AMD_MM_MID=$M; timeout 300 python -c "
import json, bench
r = bench.run_other_config('config3_32768_mixed13_23_32_fill5_fp64', steps=5)
print('config3 DBCSR_AMD_MM_MID=$M', r.get('ms_per_step'), r.get('value'))" 2>/dev/null | grep config3 ) >> $O/config3.txt
done
cat $O/config3.txt
