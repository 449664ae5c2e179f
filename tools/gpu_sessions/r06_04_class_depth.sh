#!/bin/bash
# round 6, GPU session 4: the (m, n) class kernels with 2-3 products in flight per wave (mm_exact.h: cblock_f64_classes_deep): parity of the
# mixed-size tests, then BASELINE config 3 with the depth forced to 1 (round 2's body), 2, 3 and automatic, same box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s04; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_multiply.py tests/test_gpu_plan_reuse.py -q -x 2>&1 | grep -v "$F" | tail -8 > $O/pytest_classes.txt
tail -4 $O/pytest_classes.txt
for D in 1 0 2 3 1 0; do
  ( [ $D != 0 ] && export DBCSR_AMD_MM_CLASS_DEPTH=$D; DBCSR_AMD_MM_VERBOSE=1 timeout 300 python -c "
import json, bench
r = bench.run_other_config('config3_32768_mixed13_23_32_fill5_fp64', steps=5)
r['label'] = 'depth_$D'
print(json.dumps(r))" 2> $O/err_$D.txt | grep -v "$F" | tail -1 ) >> $O/config3.jsonl
done
python3 -c "
import json
for l in open('$O/config3.jsonl'):
    if l.startswith('{'):
        d = json.loads(l); print(d.get('label'), d.get('ms_per_step'), d.get('roofline', {}).get('kernel_ms'), d.get('value'), d.get('roofline', {}).get('frac'))
"
grep "compiled class" $O/err_0.txt | head -9
