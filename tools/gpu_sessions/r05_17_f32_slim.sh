#!/bin/bash
# round 5, GPU session 17: the fp32 direct kernel launched with LDS for its B images only when every C block has the dominant size (the
# staged fall-back's slice held the CU at 16 waves): parity, 32768^2, config 5, against DBCSR_AMD_MM_F32_DIRECT=1
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s17; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 300 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_f32_group.py tests/test_gpu_multiply.py tests/test_gpu_native_multiply.py -q -k "fp32 or f32 or group" 2>&1 | grep -v "$F" | tail -4 > $O/pytest_f32.txt; tail -2 $O/pytest_f32.txt
B='[{"label":"slim"},{"label":"regular","env":["DBCSR_AMD_MM_F32_DIRECT=1"]},{"label":"slim_ww2","env":["DBCSR_AMD_MM_WG_WAVES=2"]},{"label":"slim_16384","size":16384},{"label":"regular_16384","size":16384,"env":["DBCSR_AMD_MM_F32_DIRECT=1"]}]'
timeout 200 python tools/block_bench.py --size 32768 --mix 1,32 --fill 0.2 --dtype f32 --check --batch "$B" 2>&1 | grep -v "$F" > $O/f32_slim.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r05_s17/f32_slim.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r.get("label"), r.get("size"), r.get("kernel"), r.get("kernel_ms"), r.get("tflops_kernel"), (r.get("check") or {}).get("max_abs_diff_over_max_abs"), r.get("error"))
PY
for spec in "slim:" "regular:DBCSR_AMD_MM_F32_DIRECT=1"; do
  L=${spec%%:*}; E=${spec#*:}
  ( [ -n "$E" ] && export $E; timeout 200 python -c "
import json, bench
r = bench.run_other_config('config5_131072_32x32_fill20_fp32')
r['label'] = '$L'
print(json.dumps(r))" 2>&1 | grep -v "$F" | tail -1 ) >> $O/config5.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_s17/config5.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r.get("label"), r.get("kernel"), r.get("k_passes"), r.get("ms_per_step"), r.get("kernel_ms"), r.get("frac"))
PY
