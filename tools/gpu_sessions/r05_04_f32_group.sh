#!/bin/bash
# round 5, GPU session 4: the fp32 group kernel (a wave owns R C blocks of one column and shares B): parity, R = 2 / 3 / 4 against the
# one-wave-per-block kernel on 32768^2, config 5 at full size with 4 and 8 k passes, counters of both kernels
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s04; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 400 python -m pytest tests/test_gpu_f32_group.py tests/test_gpu_kernel_variants.py -q -k "group or fp32" 2>&1 | grep -v "$F" | tail -25 > $O/pytest_group.txt
tail -6 $O/pytest_group.txt
B='[{"label":"direct","env":["DBCSR_AMD_MM_F32_GROUP=0"]},{"label":"group_R2","env":["DBCSR_AMD_MM_F32_GROUP=2"]},{"label":"group_R3","env":["DBCSR_AMD_MM_F32_GROUP=3"]},{"label":"group_R4","env":["DBCSR_AMD_MM_F32_GROUP=4"]},'
B="$B"'{"label":"group_R4_k2","env":["DBCSR_AMD_MM_F32_GROUP=4","DBCSR_AMD_MM_KCHUNKS=2"]},{"label":"group_R4_panel64","env":["DBCSR_AMD_MM_F32_GROUP=4","DBCSR_AMD_MM_PANEL_MB=64"]},{"label":"auto"}]'
timeout 300 python tools/block_bench.py --size 32768 --mix 1,32 --fill 0.2 --dtype f32 --check --batch "$B" 2>&1 | grep -v "$F" > $O/f32_group_32768.jsonl
cut -c1-330 $O/f32_group_32768.jsonl
for spec in "group_auto:" "group_R4_k8:DBCSR_AMD_MM_KCHUNKS=8" "group_R3:DBCSR_AMD_MM_F32_GROUP=3" "group_R2:DBCSR_AMD_MM_F32_GROUP=2" "group_R4_k6:DBCSR_AMD_MM_KCHUNKS=6"; do
  L=${spec%%:*}; E=${spec#*:}
  ( [ -n "$E" ] && export $E; timeout 240 python -c "
import json, bench
r = bench.run_other_config('config5_131072_32x32_fill20_fp32')
r['label'] = '$L'
print(json.dumps(r))" 2>&1 | grep -v "$F" | tail -1 ) >> $O/config5.jsonl
done
cut -c1-420 $O/config5.jsonl
bash tools/profile_cmd.sh r05_f32_group_32768 python $PWD/tools/block_bench.py --size 32768 --mix 1,32 --fill 0.2 --dtype f32 --steps 2 > $O/prof_f32_group.txt 2>&1
cp gpurun_out/prof_r05_f32_group_32768/summary.txt $O/prof_f32_group_summary.txt
DBCSR_AMD_MM_F32_GROUP=0 bash tools/profile_cmd.sh r05_f32_direct_32768 python $PWD/tools/block_bench.py --size 32768 --mix 1,32 --fill 0.2 --dtype f32 --steps 2 > $O/prof_f32_direct.txt 2>&1
cp gpurun_out/prof_r05_f32_direct_32768/summary.txt $O/prof_f32_direct_summary.txt
tail -45 $O/prof_f32_group_summary.txt
