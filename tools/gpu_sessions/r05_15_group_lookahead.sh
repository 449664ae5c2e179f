#!/bin/bash
# round 5, GPU session 15: the fp32 group kernel with two list entries of look-ahead (the scalar load of the next entry sat between a step's
# commit and the next step's peek): parity, 32768^2, config 5
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s15; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 300 python -m pytest tests/test_gpu_f32_group.py -q 2>&1 | grep -v "$F" | tail -4 > $O/pytest_group.txt; tail -2 $O/pytest_group.txt
B='[{"label":"direct","env":["DBCSR_AMD_MM_F32_GROUP=0"]},{"label":"group_R2","env":["DBCSR_AMD_MM_F32_GROUP=2"]},{"label":"group_R3","env":["DBCSR_AMD_MM_F32_GROUP=3"]},{"label":"group_R4","env":["DBCSR_AMD_MM_F32_GROUP=4"]}]'
timeout 200 python tools/block_bench.py --size 32768 --mix 1,32 --fill 0.2 --dtype f32 --batch "$B" 2>&1 | grep -v "$F" > $O/f32_group_32768.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r05_s15/f32_group_32768.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r.get("label"), r.get("kernel"), r.get("kernel_ms"), r.get("tflops_kernel"), r.get("error"))
PY
for spec in "group_R2:DBCSR_AMD_MM_F32_GROUP=2" "group_R4:DBCSR_AMD_MM_F32_GROUP=4" "group_R3:DBCSR_AMD_MM_F32_GROUP=3"; do
  L=${spec%%:*}; E=${spec#*:}
  ( export $E; timeout 200 python -c "
import json, bench
r = bench.run_other_config('config5_131072_32x32_fill20_fp32')
r['label'] = '$L'
print(json.dumps(r))" 2>&1 | grep -v "$F" | tail -1 ) >> $O/config5.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_s15/config5.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r.get("label"), r.get("kernel"), r.get("k_passes"), r.get("ms_per_step"), r.get("kernel_ms"), r.get("frac"))
PY
