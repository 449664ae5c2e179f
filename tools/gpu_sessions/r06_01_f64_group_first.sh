#!/bin/bash
# round 6, GPU session 1: the fp64 group kernel (mm_group64.hip): parity (oracle + bitwise against the one-wave-per-block kernel), then
# config 2 (32768^2, 23 x 23, 10 %) with R = 2 ... 6 and panel sizes against the production kernel
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s01; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 600 python -m pytest tests/test_gpu_f64_group.py -q -x 2>&1 | grep -v "$F" | tail -25 > $O/pytest_group.txt
tail -12 $O/pytest_group.txt
B='[{"label":"hot","env":["DBCSR_AMD_MM_F64_GROUP=0"]}'
for R in 2 3 4 5 6; do B="$B"',{"label":"group_R'$R'","env":["DBCSR_AMD_MM_F64_GROUP='$R'"]}'; done
for R in 4 6; do for P in 128 512 1024; do B="$B"',{"label":"group_R'$R'_panel'$P'","env":["DBCSR_AMD_MM_F64_GROUP='$R'","DBCSR_AMD_MM_GROUP_PANEL_MB='$P'"]}'; done; done
B="$B"',{"label":"hot_again","env":["DBCSR_AMD_MM_F64_GROUP=0"]}]'
timeout 600 python tools/block_bench.py --size 32768 --mix 1,23 --fill 0.1 --steps 5 --check --batch "$B" 2>&1 | grep -v "$F" > $O/group_config2.jsonl
cut -c1-360 $O/group_config2.jsonl
